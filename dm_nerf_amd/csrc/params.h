// params.h -- the flat parameter vector: reference state_dict order (DM_NeRF.__init__, networks/dm_nerf.py:59-78),
// each nn.Linear as weight [out, in] row-major followed by bias [out].  Shared by the packer, the weight-gradient plan
// and the head kernels.
#pragma once
#include <stdint.h>

#include "layout.h"

namespace dmn {

struct Lin {
    int64_t w_off, b_off;
    int out, in;
    DMN_HD int64_t w(int o, int i) const { return (o < out && i >= 0 && i < in) ? w_off + (int64_t)o * in + i : -1; }
    DMN_HD int64_t b(int o) const { return o < out ? b_off + o : -1; }
};

struct Params {
    Lin mlps[8], rgb_feature, ins_feature, rgb_hidden, ins_hidden, density, ins_out, rgb_out;
    int64_t total;
};

DMN_HD inline Params make_params(int ins_num) {
    Params P{};
    int64_t o = 0;
    auto add = [&](Lin& l, int out, int in) {
        l.out = out; l.in = in; l.w_off = o; o += (int64_t)out * in; l.b_off = o; o += out;
    };
    add(P.mlps[0], W, POS_CH);
    for (int i = 1; i < 8; ++i) add(P.mlps[i], W, i == 5 ? W + POS_CH : W);
    add(P.rgb_feature, W, W);
    add(P.ins_feature, W, W);
    add(P.rgb_hidden, HW, W + DIR_CH);
    add(P.ins_hidden, HW, W);
    add(P.density, 1, W);
    add(P.ins_out, ins_num + 1, HW);
    add(P.rgb_out, 3, HW);
    P.total = o;
    return P;
}

}  // namespace dmn
