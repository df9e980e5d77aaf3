"""dm_nerf_amd.dropin: the reference's own modules load unmodified and exactly their hot-path names are rebound, before
other reference modules copy them.  Needs the reference checkout (build container only; skipped elsewhere): it is run
in a subprocess so that the reference's import side effects (anomaly detection, dm_nerf.py:5) stay out of this session."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("DMNERF_REFERENCE", "/root/reference")

PROBE = r'''
import sys
from unittest.mock import MagicMock
for mod in ("imageio", "lpips", "cv2", "skimage", "skimage.metrics", "open3d", "matplotlib", "matplotlib.pyplot", "matplotlib.cm", "h5py",
            "configargparse", "trimesh", "mcubes", "plyfile"):
    sys.modules.setdefault(mod, MagicMock())              # third-party packages of the reference's drivers that this image lacks
sys.path.insert(0, %(root)r); sys.path.insert(0, %(ref)r)
import dm_nerf_amd.dropin as D
D.install()
import networks.render, networks.helpers, networks.dm_nerf, networks.penalizer, networks.evaluator, config
import networks.tester, networks.manipulator                # reference modules that copy the names at import (tester.py:11-13)
import dm_nerf_amd.networks.render as R, dm_nerf_amd.networks.helpers as H, dm_nerf_amd.networks.dm_nerf as M
import dm_nerf_amd.networks.manipulator as MA, dm_nerf_amd.config as C, dm_nerf_amd.networks.evaluator as E
assert networks.render.dm_nerf is R.dm_nerf and networks.render.render_train is R.render_train
assert networks.tester.dm_nerf is R.dm_nerf and networks.tester.get_rays_k is H.get_rays_k and networks.tester.z_val_sample is H.z_val_sample
assert networks.helpers.sample_pdf is H.sample_pdf and networks.helpers.get_select_crop is H.get_select_crop
assert networks.manipulator.manipulator is MA.manipulator and networks.manipulator.exchanger is MA.exchanger
assert networks.manipulator.sample_pdf is H.sample_pdf
assert networks.dm_nerf.DM_NeRF is M.DM_NeRF and config.DM_NeRF is M.DM_NeRF and config.create_nerf is C.create_nerf
assert networks.evaluator.ins_criterion is E.ins_criterion
# not on the hot path: still the reference's own code
assert networks.evaluator.calculate_ap.__module__ == "networks.evaluator" and networks.helpers.get_rays.__module__ == "networks.helpers"
assert config.initial.__module__ == "config" and networks.tester.render_test.__module__ == "networks.tester"
# this package's own modules are never patched
assert not hasattr(C, "__dm_nerf_amd_patched__")
print("DROPIN_OK", sorted(networks.helpers.__dm_nerf_amd_patched__))
'''


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "networks")), reason="needs the reference checkout (build container)")
def test_reference_modules_are_rebound_without_edits():
    p = subprocess.run([sys.executable, "-c", PROBE % {"root": ROOT, "ref": REF}], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "DROPIN_OK" in p.stdout, p.stderr[-2000:]


def test_patch_table_names_exist_in_this_package():
    import dm_nerf_amd.dropin as D
    for mod, names in D.PATCHES.items():
        for name, ref in names.items():
            assert callable(D._resolve(ref)) or isinstance(D._resolve(ref), type), (mod, name)
