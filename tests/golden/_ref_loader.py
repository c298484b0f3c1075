"""Import harness for the *reference* repo (qq456cvb/CPPF2) -- used ONLY by
tests/golden/make_golden.py in the build container, where /root/reference exists.

It never copies reference source: it puts /root/reference on sys.path and
installs empty stub modules for the third-party packages the reference imports
at module scope but which are absent in this image (hydra, omegaconf, open3d,
torch_scatter, cv2 ...).  Four stubs get real semantics because hot-path
functions call them:
  * torch_scatter.scatter_add  -> zeros(dim_size).index_add_(0, index, src)
    (exact for the int64 ones it is used with, train_dino.py:204, eval.py:265)
  * pytorch_lightning.LightningModule -> torch.nn.Module
  * hydra.main -> identity decorator
  * albumentations.ImageOnlyTransform -> object
Nothing in this file travels to the GPU box as a dependency of tests: the
fixtures it helps to generate are plain .npz data.
"""
import contextlib
import importlib
import importlib.abc
import importlib.machinery
import os
import sys
import types

REF_ROOT = "/root/reference"

_STUB_ROOTS = {
    "hydra", "omegaconf", "trimesh", "albumentations", "zmq", "src_shot",
    "torchvision", "open3d", "icecream", "pytorch_lightning", "wandb", "cv2",
    "skimage", "pycocotools", "visdom", "fire", "lietorch", "torch_scatter",
    "matplotlib", "PIL", "tqdm",
}


class _Stub(types.ModuleType):
    __path__ = []

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        full = self.__name__ + "." + name
        if full in sys.modules:
            return sys.modules[full]
        # any attribute is a callable/class placeholder that is also a module
        m = _Stub(full)
        sys.modules[full] = m
        setattr(self, name, m)
        return m

    def __call__(self, *a, **k):
        # used as decorator factory (hydra.main(...)) or constructor
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return self


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    # `src_shot` (the reference's unbuilt pybind11 module; this repository has a real package of that name) is stubbed only
    # while load_reference() imports the reference
    loading = False

    def find_spec(self, fullname, path, target=None):
        root = fullname.split(".")[0]
        if root == "src_shot" and not _Finder.loading:
            return None
        if root in _STUB_ROOTS or fullname == "scipy.misc":
            try:
                # prefer a real install if there is one (PIL, tqdm, matplotlib)
                if root in ("PIL", "tqdm", "matplotlib"):
                    for f in sys.meta_path:
                        if f is self:
                            continue
                        spec = f.find_spec(fullname, path, target) if hasattr(f, "find_spec") else None
                        if spec is not None:
                            return spec
            except Exception:
                pass
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return _Stub(spec.name)

    def exec_module(self, module):
        pass


# Top-level names BOTH trees define.  The reference's `utils/` has no __init__.py (a namespace package), this repository's
# `utils/` is a regular package and therefore wins whatever the order of sys.path is; `dataset`, `eval`, `train_*` and `src_shot`
# exist at both roots too.  load_reference() imports the reference's modules with this repository's root taken OFF sys.path and
# these names purged from sys.modules, and afterwards restores both, so that `cppf2_amd` (and this repository's shims) import as
# usual while the returned namespace holds the reference's own module objects.
_SHARED_NAMES = ("utils", "dataset", "eval", "train_dino", "train_shot", "src_shot", "demo")
_REPO_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_CACHE = None


def _is_repo_root(p):
    try:
        return os.path.realpath(p or os.getcwd()) == os.path.realpath(_REPO_ROOT)
    except OSError:
        return False


def _shared(name):
    return name.split(".")[0] in _SHARED_NAMES


def load_reference():
    """Returns a namespace with the reference's hot-path callables and its modules (`ns.dataset`, `ns.eval`, `ns.train_dino`,
    `ns.train_shot`, `ns.util`) -- all of them loaded from /root/reference, checked by file path."""
    global _CACHE
    if _CACHE is not None:
        return _CACHE
    import torch
    if not any(isinstance(f, _Finder) for f in sys.meta_path):
        sys.meta_path.insert(0, _Finder())
    saved_modules = {k: sys.modules.pop(k) for k in list(sys.modules) if _shared(k)}
    saved_path = list(sys.path)
    sys.path[:] = [REF_ROOT] + [p for p in sys.path if p != REF_ROOT and not _is_repo_root(p)]
    importlib.invalidate_caches()
    _Finder.loading = True
    ref_modules = {}
    try:
        ns = _import_reference(torch)
    finally:
        _Finder.loading = False
        for k in [k for k in sys.modules if _shared(k)]:
            ref_modules[k] = sys.modules.pop(k)
        sys.modules.update(saved_modules)
        sys.path[:] = saved_path
        importlib.invalidate_caches()
    ns._modules = ref_modules
    _CACHE = ns
    return ns


@contextlib.contextmanager
def reference_modules():
    """Puts the reference's module objects back under their names in sys.modules for the duration of the block: needed where
    the reference pickles its own functions by qualified name (multiprocessing in compute_degree_cm_mAP, utils/util.py:2771-2779)."""
    ns = load_reference()
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if _shared(k)}
    sys.modules.update(ns._modules)
    try:
        yield ns
    finally:
        for k in [k for k in sys.modules if _shared(k)]:
            del sys.modules[k]
        sys.modules.update(saved)


def _import_reference(torch):
    import pytorch_lightning as pl
    pl.LightningModule = torch.nn.Module
    import hydra
    hydra.main = lambda *a, **k: (lambda fn: fn)
    import albumentations as A
    A.ImageOnlyTransform = object
    import torch_scatter

    def scatter_add(src, index, dim=-1, out=None, dim_size=None):
        if dim_size is None:
            dim_size = int(index.max()) + 1
        res = torch.zeros((int(dim_size),), dtype=src.dtype, device=src.device)
        return res.index_add_(0, index, src)
    torch_scatter.scatter_add = scatter_add

    import train_dino
    import train_shot
    import dataset
    import utils.util as util

    import eval as ref_eval
    for m in (train_dino, train_shot, dataset, util, ref_eval):
        assert os.path.realpath(m.__file__).startswith(os.path.realpath(REF_ROOT) + os.sep), \
            "%s resolved to %s, not to the reference" % (m.__name__, m.__file__)

    ns = types.SimpleNamespace()
    ns.train_dino, ns.train_shot, ns.dataset, ns.eval = train_dino, train_shot, dataset, ref_eval
    ns.vote_center = train_dino.vote_center
    ns.vote_rotation = train_dino.vote_rotation
    ns.generate_target_pairs = dataset.generate_target_pairs
    ns.BeyondCPPFDINO = train_dino.BeyondCPPF
    ns.BeyondCPPFSHOT = train_shot.BeyondCPPF
    ns.fibonacci_sphere = util.fibonacci_sphere
    ns.real2prob = util.real2prob
    ns.prob2real = util.prob2real
    ns.backproject = util.backproject
    ns.util = util

    # eval.get_topk_dir hard-codes .cuda()/device='cuda' (eval.py:38-39):
    # run it on CPU by making .cuda() the identity and dropping device kwargs.
    _zeros = torch.zeros

    def get_topk_dir_cpu(*a, **k):
        orig_cuda = torch.Tensor.cuda
        torch.Tensor.cuda = lambda self, *aa, **kk: self
        torch.zeros = lambda *aa, **kk: _zeros(*aa, **{x: y for x, y in kk.items() if x != "device"})
        try:
            return ref_eval.get_topk_dir(*a, **k)
        finally:
            torch.Tensor.cuda = orig_cuda
            torch.zeros = _zeros
    ns.get_topk_dir = get_topk_dir_cpu
    return ns
