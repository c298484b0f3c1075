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
import importlib.abc
import importlib.machinery
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
    def find_spec(self, fullname, path, target=None):
        root = fullname.split(".")[0]
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


def load_reference():
    """Returns a namespace with the reference's hot-path callables."""
    import torch
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    if not any(isinstance(f, _Finder) for f in sys.meta_path):
        sys.meta_path.insert(0, _Finder())

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

    ns = types.SimpleNamespace()
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
    import eval as ref_eval
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
