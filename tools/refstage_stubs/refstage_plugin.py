"""pytest plugin of tools/reference_on_hip.py (``-p refstage_plugin``): third-party modules the reference imports at module
level and the hot path never needs (trimesh, lxml) are stubbed; ``warp`` resolves to the stand-in with the device shim
(.refstage/_stubs/warp); ``yourdfpy`` to the xml.etree stand-in; the Warp-side hooks of INTEGRATION.md are installed as
soon as their modules are imported; a report of what ran where is written at the end."""
import importlib.abc
import importlib.machinery
import json
import os
import sys
from unittest.mock import MagicMock

import hip_hooks


class _Stub(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    ROOTS = {"trimesh", "lxml"}

    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in self.ROOTS:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)

    def create_module(self, spec):
        m = MagicMock(name=spec.name)
        m.__path__, m.__name__, m.__spec__, m.__loader__ = [], spec.name, spec, self
        return m

    def exec_module(self, module):
        pass


sys.meta_path.append(_Stub())
_patched = []


def pytest_collection_finish(session):
    _patched.extend(hip_hooks.install())


def pytest_runtest_setup(item):
    _patched.extend(hip_hooks.install())


def pytest_runtest_call(item):
    _patched.extend(hip_hooks.install())


def pytest_sessionfinish(session, exitstatus):
    out = os.environ.get("REFSTAGE_REPORT")
    if not out:
        return
    import warp

    loaded = [l.split()[-1] for l in open("/proc/self/maps") if "libcurobo_hip" in l]
    rec = {"hooks_installed": sorted(set(_patched)), "hip_entry_points_reached": hip_hooks.CALLS,
           "warp_stand_in_launches_on_host": getattr(warp, "LAUNCH_LOG", {}), "libcurobo_hip_mapped": sorted(set(loaded))}
    with open(out, "w") as fh:
        json.dump(rec, fh, indent=1)
