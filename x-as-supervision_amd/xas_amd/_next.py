"""Fall-through to the reference checkout for names the mirror does not replace.

The mirror (`x-as-supervision_amd/`) sits in FRONT of the reference checkout on PYTHONPATH and only re-implements what is
on the training / evaluation hot path.  Everything else (`train_util.basic_data`, the dataset classes of `human_utils.dataset`,
the cv2 loader helpers, the plotting helpers of `eval_utils`, dead-code helpers of `modules.util`) must keep resolving to the
reference's own files (reference: train_util.py:9-13,16-106; human_utils/dataloader/dataloader.py:8-13; eval.py:20-24).  Two pieces:

* every mirrored package `__init__` extends its `__path__` over the later `sys.path` entries (`pkgutil.extend_path`), so a
  sub-module the mirror does not have (`human_utils.dataset`, `human_utils.dataloader.dataloader`,
  `modules.base_losses.integral`, `human_utils.common.visualization`) is found in the reference checkout;
* a mirrored MODULE that replaces only some names of the reference module installs `__getattr__ = fallthrough(__name__, __file__)`:
  a name it does not define is looked up - lazily, on first use - in the next module of the same dotted name on the path.
"""
import importlib.util
import os
import sys


def _candidates(name):
    """Directories in which the module `name` may live, in path order."""
    parent, _, leaf = name.rpartition('.')
    if parent:
        pkg = sys.modules.get(parent) or importlib.import_module(parent)
        dirs = list(getattr(pkg, '__path__', []))
    else:
        dirs = [d or os.getcwd() for d in sys.path]
    return leaf, dirs


def next_module(name, own_file):
    """The next module called `name` on the path after the one in `own_file` (loaded once under `<name>.__ref__`), or None."""
    key = name + '.__ref__'
    if key in sys.modules:
        return sys.modules[key]
    leaf, dirs = _candidates(name)
    own = os.path.realpath(own_file)
    for d in dirs:
        for cand in (os.path.join(d, leaf + '.py'), os.path.join(d, leaf, '__init__.py')):
            if os.path.isfile(cand) and os.path.realpath(cand) != own:
                spec = importlib.util.spec_from_file_location(key, cand)
                mod = importlib.util.module_from_spec(spec)
                sys.modules[key] = mod
                try:
                    spec.loader.exec_module(mod)
                except BaseException:
                    del sys.modules[key]
                    raise
                return mod
    return None


def fallthrough(name, own_file):
    """Module-level `__getattr__` (PEP 562) that resolves names missing from the mirror in the reference's module."""

    def __getattr__(attr):
        if attr.startswith('__') and attr.endswith('__'):
            raise AttributeError(attr)
        ref = next_module(name, own_file)
        if ref is None:
            raise AttributeError(
                f"module '{name}' (MI355X mirror) does not replace '{attr}' and no reference module '{name}' follows it on "
                f"sys.path - put the X-as-Supervision checkout on PYTHONPATH behind the mirror")
        try:
            return getattr(ref, attr)
        except AttributeError:
            raise AttributeError(f"neither the mirror nor the reference module '{name}' defines '{attr}'") from None

    return __getattr__
