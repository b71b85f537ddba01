"""Import the reference's OWN hot-path modules read-only from /root/reference (test infra).

Works only in the build container (the GPU box has no /root/reference); used by
``oracle/gen_golden.py`` to produce the committed fixtures under tests/golden/
and by the optional ``ref``-marked pinning tests.  Nothing is copied: the
reference files are imported where they lie.

Shims needed because of packages missing from this image:
  * ``pointnet2._ext``  -> oracle.pn2 (CPU restatement of the CUDA-only extension)
  * ``timm``            -> oracle.timm_standin (un-vendored, unpinned third party)
  * torchvision / pytorch_lightning / hydra / trimesh / ruamel / cv2 / imageio ...
                        -> inert attribute-absorbing stubs (never executed on the
                           scoring path; they only satisfy ``import`` lines)
"""
import importlib
import importlib.util
import os
import sys
import types

REF_ROOT = os.environ.get("S6D_REFERENCE_ROOT", "/root/reference")
PEM = os.path.join(REF_ROOT, "SAM-6D", "Pose_Estimation_Model")
ISM = os.path.join(REF_ROOT, "SAM-6D", "Instance_Segmentation_Model")


def available():
    return os.path.isdir(PEM) and os.path.isdir(ISM)


class _Absorb:
    """Callable/attribute sink used for stubbed third-party symbols."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Absorb()

    def __getattr__(self, name):
        return _Absorb()


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        if name and name[0].isupper():  # looks like a class: must be subclassable
            cls = type(name, (object,), {})
            setattr(self, name, cls)
            return cls
        return _Absorb()


def _stub(name):
    parts = name.split(".")
    for i in range(1, len(parts) + 1):
        sub = ".".join(parts[:i])
        if sub not in sys.modules:
            m = _StubModule(sub)
            m.__path__ = []  # behave like a package
            sys.modules[sub] = m
            if i > 1:
                setattr(sys.modules[".".join(parts[: i - 1])], parts[i - 1], m)


def _try_import(name):
    try:
        importlib.import_module(name)
        return True
    except Exception:
        return False


def pem():
    """Returns a namespace with the reference PEM modules (CPU-runnable)."""
    assert available(), "reference tree not present"
    from . import pn2, timm_standin

    timm_standin.install()
    pkg = types.ModuleType("pointnet2")
    pkg.__path__ = []
    pkg._ext = pn2
    sys.modules.setdefault("pointnet2", pkg)
    sys.modules.setdefault("pointnet2._ext", pn2)
    for p in (os.path.join(PEM, "model"), os.path.join(PEM, "utils"), os.path.join(PEM, "model", "pointnet2")):
        if p not in sys.path:
            sys.path.insert(0, p)
    ns = types.SimpleNamespace()
    for name in ("pointnet2_utils", "pytorch_utils", "model_utils", "transformer", "feature_extraction",
                 "coarse_point_matching", "fine_point_matching", "pose_estimation_model"):
        setattr(ns, name, importlib.import_module(name))
    return ns


def pem_cfg():
    """``model:`` node of Pose_Estimation_Model/config/base.yaml as attribute dicts
    (gorilla.Config is absent; attribute access is all the constructors use)."""
    import yaml

    class AD(dict):
        __getattr__ = dict.__getitem__

    def conv(o):
        return AD({k: conv(v) for k, v in o.items()}) if isinstance(o, dict) else o

    with open(os.path.join(PEM, "config", "base.yaml")) as f:
        cfg = conv(yaml.safe_load(f))
    cfg.model.feature_extraction["pretrained"] = False  # MAE checkpoint needs the network
    return cfg


def _load_by_path(modname, path, package=None):
    spec = importlib.util.spec_from_file_location(modname, path)
    mod = importlib.util.module_from_spec(spec)
    if package:
        mod.__package__ = package
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


def sam_encoder():
    """segment_anything/modeling/{common,image_encoder}.py loaded by file path, bypassing
    segment_anything/__init__.py (which imports torchvision)."""
    assert available()
    base = os.path.join(ISM, "segment_anything", "modeling")
    pkgname = "_s6d_ref_sa_modeling"
    if pkgname + ".image_encoder" in sys.modules:
        return sys.modules[pkgname + ".image_encoder"]
    pkg = types.ModuleType(pkgname)
    pkg.__path__ = [base]
    sys.modules[pkgname] = pkg
    _load_by_path(pkgname + ".common", os.path.join(base, "common.py"), pkgname)
    return _load_by_path(pkgname + ".image_encoder", os.path.join(base, "image_encoder.py"), pkgname)


def pem_data_utils():
    """Pose_Estimation_Model/utils/data_utils.py loaded by file path (imageio / cv2 / PIL stubbed: the geometry helpers
    used for the goldens are pure numpy)."""
    assert available()
    for name in ("imageio", "cv2", "PIL", "PIL.Image"):
        root = name.split(".")[0]
        if root not in sys.modules and not _try_import(root):
            _stub(name)
    return sys.modules.get("_s6d_ref_pem_data_utils") or _load_by_path(
        "_s6d_ref_pem_data_utils", os.path.join(PEM, "utils", "data_utils.py"))


def sam_decoder():
    """segment_anything/modeling/{common,prompt_encoder,transformer,mask_decoder}.py loaded by file path (same
    package trick as sam_encoder()).  Returns a namespace with PromptEncoder, MaskDecoder, TwoWayTransformer."""
    assert available()
    base = os.path.join(ISM, "segment_anything", "modeling")
    pkgname = "_s6d_ref_sa_modeling"
    sam_encoder()                                                    # creates the package + common
    ns = types.SimpleNamespace()
    for name in ("prompt_encoder", "transformer", "mask_decoder"):
        full = pkgname + "." + name
        ns.__dict__[name] = sys.modules.get(full) or _load_by_path(full, os.path.join(base, name + ".py"), pkgname)
    ns.amg = sys.modules.get("_s6d_ref_sa_amg") or _load_by_path(
        "_s6d_ref_sa_amg", os.path.join(ISM, "segment_anything", "utils", "amg.py"))
    for name in ("torchvision", "torchvision.transforms", "torchvision.transforms.functional"):
        if name.split(".")[0] not in sys.modules and not _try_import(name.split(".")[0]):
            _stub(name)
        elif isinstance(sys.modules.get(name.split(".")[0]), _StubModule):
            _stub(name)
    ns.transforms = sys.modules.get("_s6d_ref_sa_transforms") or _load_by_path(
        "_s6d_ref_sa_transforms", os.path.join(ISM, "segment_anything", "utils", "transforms.py"))
    ns.PromptEncoder = ns.prompt_encoder.PromptEncoder
    ns.MaskDecoder = ns.mask_decoder.MaskDecoder
    ns.TwoWayTransformer = ns.transformer.TwoWayTransformer
    return ns


def sam_builder():
    """segment_anything/build_sam.py loaded by file path: a synthetic parent package whose ``.modeling`` exposes the
    reference's own classes (modeling/sam.py is loaded into the same synthetic modeling package).  Returns the module
    (sam_model_registry, build_sam_vit_h ...)."""
    ns = sam_decoder()
    mp = "_s6d_ref_sa_modeling"
    base = os.path.join(ISM, "segment_anything")
    sam_mod = sys.modules.get(mp + ".sam") or _load_by_path(mp + ".sam", os.path.join(base, "modeling", "sam.py"), mp)
    pkg = sys.modules[mp]
    pkg.ImageEncoderViT = sys.modules[mp + ".image_encoder"].ImageEncoderViT
    pkg.MaskDecoder, pkg.PromptEncoder, pkg.TwoWayTransformer, pkg.Sam = ns.MaskDecoder, ns.PromptEncoder, ns.TwoWayTransformer, sam_mod.Sam
    top = "_s6d_ref_sa"
    if top not in sys.modules:
        t = types.ModuleType(top)
        t.__path__ = [base]
        t.modeling = pkg
        sys.modules[top] = t
        sys.modules[top + ".modeling"] = pkg
    return sys.modules.get(top + ".build_sam") or _load_by_path(top + ".build_sam", os.path.join(base, "build_sam.py"), top)


def ism():
    """Reference ISM scoring code: model.loss classes, detector scoring methods,
    compute_iou and the masked-depth translation helper."""
    assert available()
    for name in ("torchvision", "torchvision.ops", "torchvision.ops.boxes", "torchvision.transforms",
                 "torchvision.transforms.functional", "torchvision.utils", "pytorch_lightning",
                 "hydra", "hydra.utils", "trimesh", "ruamel", "ruamel.yaml", "cv2", "imageio",
                 "pycocotools", "pycocotools.mask", "omegaconf", "distinctipy", "skimage", "skimage.feature",
                 "skimage.filters", "skimage.measure", "pandas"):
        root = name.split(".")[0]
        real = (root in sys.modules and not isinstance(sys.modules[root], _StubModule)) or (
            root not in sys.modules and _try_import(root))
        if not real:
            _stub(name)
    import torch.nn as nn

    sys.modules["pytorch_lightning"].LightningModule = nn.Module
    if ISM not in sys.path:
        sys.path.insert(0, ISM)
    # the ISM tree has top-level packages called `model` and `utils`
    for clash in ("model", "utils"):
        m = sys.modules.get(clash)
        if m is not None and not getattr(m, "__file__", "").startswith(ISM):
            raise RuntimeError(f"module name clash on '{clash}'")
    ns = types.SimpleNamespace()
    ns.loss = importlib.import_module("model.loss")
    ns.detector = importlib.import_module("model.detector")
    ns.bbox_utils = importlib.import_module("utils.bbox_utils")
    ns.trimesh_utils = importlib.import_module("utils.trimesh_utils")
    return ns
