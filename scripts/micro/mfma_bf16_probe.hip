// Probe (GPU box): operand layout of v_mfma_f32_32x32x16_bf16 on gfx950.
// Hypothesis: A[i][k]: lane l holds row i = l % 32, k = 8 (l / 32) + q (q = 0..7, ascending in its 8 bf16);
//             B[k][j]: lane l holds col j = l % 32, k = 8 (l / 32) + q;  C[i][j]: lane l, register r: j = l % 32,
//             i = (r & 3) + 8 (r >> 2) + 4 (l / 32)   (as for v_mfma_f32_32x32x2_f32).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ void probe(const float* A, const float* B, float* C) {   // A [32][16], B [16][32] row-major, C [32][32]
    const int l = threadIdx.x, i = l % 32, h = l / 32;
    bf16x8 a, b;
    for (int q = 0; q < 8; ++q) { a[q] = (__bf16)A[i * 16 + 8 * h + q]; b[q] = (__bf16)B[(8 * h + q) * 32 + i]; }
    f32x16 c = (f32x16)(0.f);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = c[r];
}

int main() {
    float hA[512], hB[512], hC[1024], ref[1024];
    for (int i = 0; i < 512; ++i) { hA[i] = (float)((i * 7 + 3) % 13 - 6); hB[i] = (float)((i * 5 + 1) % 11 - 5); }
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int k = 0; k < 16; ++k) s += hA[i * 16 + k] * hB[k * 32 + j]; ref[i * 32 + j] = s; }
    float *dA, *dB, *dC;
    (void)hipMalloc(&dA, 2048); (void)hipMalloc(&dB, 2048); (void)hipMalloc(&dC, 4096);
    (void)hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dA, dB, dC);
    (void)hipMemcpy(hC, dC, 4096, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 1024; ++i) bad += hC[i] != ref[i];
    printf("v_mfma_f32_32x32x16_bf16 layout hypothesis: %s (%d of 1024 mismatches)\n", bad ? "WRONG" : "confirmed", bad);
    return 0;
}
