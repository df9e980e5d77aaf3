"""Zero-edit drop-in: run the reference's own scripts with the hot path replaced.

    python -m dm_nerf_amd.dropin train_dmsr.py --config configs/dmsr/train/study.txt        # from the reference checkout
    # or, as the first line of a script:   import dm_nerf_amd.dropin as d; d.install()

The reference (vLAR-group/DM-NeRF) has no plugin interface: its boundary is Python callables imported by name
(``from networks.render import dm_nerf``, ``from config import initial, create_nerf``, ...).  ``install()``
puts a finder in front of ``sys.meta_path`` that lets the reference's modules load as they are and then rebinds, inside
them, exactly the hot-path names to this package's implementations -- before any later ``from networks.x import y`` in
another reference module (tester.py, manipulator.py, the train / test scripts) copies them.  Everything else in those
modules (argument parsing, data loaders, metrics, image output) stays the reference's own code.

What is rebound (reference file:line -> replacement):
"""
import importlib
import importlib.abc
import importlib.util
import os
import runpy
import sys

# module -> {name: "module_in_this_package:attribute"}; resolved lazily so that installing costs nothing until a module loads
PATCHES = {
    "networks.render": {"dm_nerf": "networks.render:dm_nerf",                    # render.py:31-96
                        "render_train": "networks.render:render_train"},        # render.py:6-28
    "networks.dm_nerf": {"Embedder": "networks.dm_nerf:Embedder",               # dm_nerf.py:8-38
                         "get_embedder": "networks.dm_nerf:get_embedder",       # dm_nerf.py:41-55
                         "DM_NeRF": "networks.dm_nerf:DM_NeRF"},                # dm_nerf.py:58-106
    "networks.helpers": {"get_rays_k": "networks.helpers:get_rays_k",           # helpers.py:50-61
                         "get_select_crop": "networks.helpers:get_select_crop", # helpers.py:64-95
                         "get_select_full": "networks.helpers:get_select_full", # helpers.py:99-111
                         "z_val_sample": "networks.helpers:z_val_sample",       # helpers.py:114-119
                         "sample_pdf": "networks.helpers:sample_pdf"},          # helpers.py:123-155
    "networks.penalizer": {"emptiness_penalizer": "networks.penalizer:emptiness_penalizer",   # penalizer.py:5-55
                           "ins_penalizer": "networks.penalizer:ins_penalizer"},              # penalizer.py:58-62
    "networks.evaluator": {"ins_criterion": "networks.evaluator:ins_criterion",  # evaluator.py:19-37 (+ hungarian :41-74)
                           "img2mse": "networks.evaluator:img2mse",              # evaluator.py:11
                           "mse2psnr": "networks.evaluator:mse2psnr"},           # evaluator.py:15
    "networks.manipulator": {"exchanger": "networks.manipulator:exchanger",                       # manipulator.py:18-83
                             "manipulator_render": "networks.manipulator:manipulator_render",     # :86-105
                             "manipulator_nerf": "networks.manipulator:manipulator_nerf",         # :108-134
                             "manipulator": "networks.manipulator:manipulator"},                  # :137-205
    "config": {"create_nerf": "config:create_nerf"},                                              # config.py:126-138
}
__doc__ += "\n".join(f"    {m}: {', '.join(sorted(v))}" for m, v in PATCHES.items()) + "\n"


def _resolve(ref):
    mod, attr = ref.split(":")
    return getattr(importlib.import_module("dm_nerf_amd." + mod), attr)


def apply_patches(module):
    """Rebind the hot-path names of one loaded reference module (idempotent).  Returns the names rebound."""
    done = []
    for name, ref in PATCHES.get(module.__name__, {}).items():
        setattr(module, name, _resolve(ref))
        done.append(name)
    module.__dm_nerf_amd_patched__ = tuple(done)
    return done


class _Loader(importlib.abc.Loader):
    """The reference's own loader with ONE change: after the module body has run, the hot-path names are rebound.  Everything
    else -- ``get_source`` / ``get_code`` / ``get_filename`` / ``is_package`` / ``get_data``, which ``inspect.getsource``,
    ``linecache`` tracebacks, ``pkgutil`` and ``runpy`` ask the module's ``__loader__`` for -- is the inner loader's."""

    def __init__(self, inner):
        self.inner = inner

    def create_module(self, spec):
        return self.inner.create_module(spec)

    def exec_module(self, module):
        self.inner.exec_module(module)          # the reference's module body runs unmodified ...
        apply_patches(module)                   # ... then the hot-path names point here

    def __getattr__(self, name):                # (only called for attributes this class does not define)
        return getattr(self.inner, name)


def _looks_like_reference(origin):
    """Is this file part of a DM-NeRF checkout?  A top-level ``config.py`` or ``networks/`` package of some OTHER project on
    sys.path must not be patched: the reference's layout has ``networks/render.py`` + ``networks/dm_nerf.py`` next to ``config.py``."""
    d = os.path.dirname(os.path.abspath(origin))
    root = os.path.dirname(d) if os.path.basename(d) == "networks" else d
    return all(os.path.isfile(os.path.join(root, *rel)) for rel in (("config.py",), ("networks", "render.py"), ("networks", "dm_nerf.py")))


class _Finder(importlib.abc.MetaPathFinder):
    def find_spec(self, name, path=None, target=None):
        if name not in PATCHES:
            return None
        for f in sys.meta_path:
            if f is self or not hasattr(f, "find_spec"):
                continue
            spec = f.find_spec(name, path, target)
            if spec is not None and spec.loader is not None:
                # never patch this package's own modules (e.g. dm_nerf_amd.config is found as "dm_nerf_amd.config", not "config")
                origin = os.path.abspath(spec.origin or "")
                if origin.startswith(os.path.dirname(os.path.abspath(__file__)) + os.sep):
                    return None
                if not spec.origin or not _looks_like_reference(spec.origin):
                    return None                                      # some other project's module of the same name
                spec.loader = _Loader(spec.loader)
                return spec
        return None


_installed = None


def install():
    """Idempotent.  Also patches reference modules that were imported before the call."""
    global _installed
    if _installed is None:
        _installed = _Finder()
        sys.meta_path.insert(0, _installed)
    for name in PATCHES:
        m = sys.modules.get(name)
        if m is not None and not getattr(m, "__dm_nerf_amd_patched__", None) and not (getattr(m, "__name__", "").startswith("dm_nerf_amd")):
            apply_patches(m)
    return _installed


def uninstall():
    global _installed
    if _installed is not None and _installed in sys.meta_path:
        sys.meta_path.remove(_installed)
    _installed = None


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        raise SystemExit("usage: python -m dm_nerf_amd.dropin <reference script.py> [its arguments ...]")
    script = os.path.abspath(argv[0])
    sys.argv = [script] + argv[1:]
    sys.path.insert(0, os.path.dirname(script))          # what `python script.py` does
    install()
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
