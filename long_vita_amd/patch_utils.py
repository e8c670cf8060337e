"""Import-time monkey-patch manager with the semantics of the reference's plugin surface
(M/patch_utils.py: `MindSpeedPatchesManager.register_patch(orig_func_name, new_func,
force_patch=False, create_dummy=False)` / `apply_patches()`):

  * a target is a dotted path "pkg.mod.attr" or "pkg.mod.Class.attr";
  * a replacement whose __name__ ends in "wrapper" or "decorator" is applied as
    `new = replacement(original)`, several may stack in registration order; anything else replaces
    the target outright, and registering a second outright replacement raises unless force_patch;
  * on apply, the new object is also written into every already-imported module that holds the
    ORIGINAL object under the same attribute name (identity match), so `from x import f` copies
    made before patching are redirected too;
  * create_dummy=True fabricates missing modules / attributes so that optional dependencies can be
    patched in before they exist.

One deliberate difference: the identity scan reads each module's `__dict__` instead of calling `hasattr` on it.  The
reference's `hasattr` (:65-70) makes lazy modules (transformers' `_LazyModule`) import optional dependencies and raises
`ModuleNotFoundError` out of `apply_patches` when one is absent; for real module attributes the two agree
(tests/golden/patch_manager.pt, made in a fresh interpreter for that reason).
"""
from __future__ import annotations

import importlib
import sys
import types
from typing import Callable, Dict, List, Optional


def _is_wrapper(obj) -> bool:
    name = getattr(obj, "__name__", "")
    return name.endswith("wrapper") or name.endswith("decorator")


def _missing(name: str) -> Callable:
    def dummy_function(*args, **kwargs):
        raise RuntimeError(f"function {name} no exist")
    return dummy_function


class Patch:
    def __init__(self, target: str, replacement, create_dummy: bool):
        self.target = target
        self.owner_path, _, self.attr = target.rpartition(".")
        if not self.owner_path:                    # a bare module name
            self.owner_path, self.attr = target, None
        self.create_dummy = create_dummy
        self.replacement = None
        self.wrappers: List[Callable] = []
        self.applied = False
        self.set_patch_func(replacement if replacement is not None else _missing(target))

    def set_patch_func(self, new_func, force_patch: bool = False) -> None:
        if _is_wrapper(new_func):
            self.wrappers.append(new_func)
        else:
            if self.replacement is not None and not force_patch:
                raise RuntimeError(f"the patch of {self.attr} exist !")
            self.replacement = new_func
        self.applied = False

    # -- resolution ---------------------------------------------------------------------------
    def _resolve_owner(self):
        """Import as much of owner_path as is a module, then walk attributes (classes)."""
        parts = self.owner_path.split(".")
        obj = None
        for i in range(len(parts), 0, -1):
            name = ".".join(parts[:i])
            try:
                obj = importlib.import_module(name)
            except ModuleNotFoundError:
                continue
            rest = parts[i:]
            break
        else:
            if not self.create_dummy:
                raise ModuleNotFoundError(f"No module named '{parts[0]}'")
            obj, rest = None, parts
        if obj is None or (rest and not hasattr(obj, rest[0])):
            if not self.create_dummy:
                if obj is None:
                    raise ModuleNotFoundError(self.owner_path)
                raise ModuleNotFoundError(f"{self.owner_path}: no attribute {rest[0]}")
            # fabricate the missing module chain
            built = [] if obj is None else self.owner_path.split(".")[: len(parts) - len(rest)]
            for seg in rest:
                built.append(seg)
                name = ".".join(built)
                mod = types.ModuleType(name)
                mod.__file__ = "mindspeed.dummy_module.py"               # the reference's marker string (:85)
                sys.modules[name] = mod
                if obj is not None:
                    setattr(obj, seg, mod)
                obj = mod
            return obj
        for seg in rest:
            obj = getattr(obj, seg)
        return obj

    def apply_patch(self) -> None:
        if self.applied:
            return
        owner = self._resolve_owner()
        original = None
        if self.attr is not None:
            if hasattr(owner, self.attr):
                original = getattr(owner, self.attr)
            elif self.create_dummy:
                original = _missing(self.target)
            elif isinstance(owner, types.ModuleType):
                original = None
            else:
                raise RuntimeError(f"no exist {self.attr} of {owner}")
        # like the reference (:54-71) the patch function is cumulative: a patch that is re-applied after another wrapper
        # was registered wraps the ALREADY wrapped function with every registered wrapper again
        if self.replacement is None:
            self.replacement = original
        for w in self.wrappers:
            self.replacement = w(self.replacement)
        new = self.replacement
        if self.attr is not None:
            setattr(owner, self.attr, new)
            # identity match on the ORIGINAL object — None included (the reference compares id()s, :65-70), so a module
            # that holds `attr = None` under the same name is redirected as well
            for mod in list(sys.modules.values()):
                if mod is None:
                    continue
                try:
                    d = mod.__dict__ if hasattr(mod, "__dict__") else {}
                    if self.attr in d and d[self.attr] is original:
                        setattr(mod, self.attr, new)
                except Exception:  # pragma: no cover - exotic module objects
                    continue
        self.applied = True


class MindSpeedPatchesManager:
    """Name kept so that `from long_vita_megatron.patch_utils import MindSpeedPatchesManager as aspm`
    call sites read the same."""
    patches_info: Dict[str, Patch] = {}

    @staticmethod
    def register_patch(orig_func_name: str, new_func=None, force_patch: bool = False, create_dummy: bool = False):
        info = MindSpeedPatchesManager.patches_info
        if orig_func_name not in info:
            info[orig_func_name] = Patch(orig_func_name, new_func, create_dummy)
        else:
            info[orig_func_name].set_patch_func(new_func, force_patch)

    @staticmethod
    def apply_patches():
        for patch in MindSpeedPatchesManager.patches_info.values():
            patch.apply_patch()
