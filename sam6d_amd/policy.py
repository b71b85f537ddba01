"""Precision / path policy of the library: WHICH dtype a model runs in and WHICH optional path it takes -- one in-memory object
instead of `os.environ` reads inside forward() (~25 S6D_* variables were read on every call).

The reference configures precision through its cfg / Lightning trainer (`Instance_Segmentation_Model/configs/machine/trainer/
local.yaml:9`, `precision: 16`); here the equivalent is a `PrecisionPolicy`:

* the process starts with `PrecisionPolicy.from_env()`: the `S6D_*` variables are the DEFAULT SOURCE, read once at import;
* callers pick another one explicitly -- `policy.set(pem_vit_dtype="fp16")`, `with policy.use(sam_gemm="fp8"): ...`,
  `policy.set_policy(PrecisionPolicy.benched())` (what bench.py runs) -- and `policy.reload()` re-reads the environment for code
  that still speaks through variables (the test-suite's monkeypatch, tools/);
* forward paths read attributes of `policy.current()` -- never the environment.

Strict mode (`strict=True` / `S6D_STRICT=1`): every module has the shape "if the kernel applies: kernel, else
the reference's torch statements".  `guard(site, **conditions)` is that `if`: it returns whether all conditions hold, and when
they do not ON A CUDA TENSOR it records (site, first failed condition) in `library_branch_hits()` -- and raises under strict
mode, naming the failed guard.  The parity tests cannot see such a fall-back (the library branch is the reference's arithmetic);
this makes it observable.
"""
import contextlib
import dataclasses
import os

_ENV = {   # field: (variable, default, type)
    "sam_dtype": ("S6D_SAM_DTYPE", "bf16", str),                 # SAM ViT-H image encoder: bf16 | fp32
    "sam_gemm": ("S6D_SAM_GEMM", "bf16", str),                   # its LayerNorm-fed GEMMs: bf16 | fp8 | fp8mx (BASELINE configs[4])
    "sam_decoder_dtype": ("S6D_SAM_DECODER_DTYPE", "bf16", str),  # prompt encoder + mask decoder: bf16 | fp32
    "sam_mlp_rows": ("S6D_SAM_MLP_ROWS", 0, int),                # > 0: MLP row slab of the library path
    "samdec_gemm": ("S6D_SAMDEC_GEMM", "kernel", str),           # mask decoder Linears: kernel | library
    "samdec_t2i": ("S6D_SAMDEC_T2I", "raw", str),                # token-to-image attention on raw (un-projected) keys
    "amg_graph": ("S6D_AMG_GRAPH", "1", str),                    # hipGraph replay of the mask generator's decoder batches
    "amg_profile": ("S6D_AMG_PROFILE", "", str),
    "dino_dtype": ("S6D_DINO_DTYPE", "bf16", str),               # DINOv2 ViT-L descriptor model: bf16 | fp32
    "dino_gemm": ("S6D_DINO_GEMM", "bf16", str),                 # bf16 | fp8 | fp8mx
    "pem_vit_dtype": ("S6D_PEM_VIT_DTYPE", "fp32", str),         # PEM ViT-B feature extractor: fp32 (reference precision) | fp16 | bf16
    "pem_geo_dtype": ("S6D_PEM_GEO_DTYPE", "fp32", str),         # geometric embedding: fp32 | fp16 (opt-in: flips 1 coarse hypothesis in 32)
    "pem_f16_guard": ("S6D_PEM_F16_GUARD", "1", str),            # re-run in fp32 when the half extractor overflows
    "pem_graph": ("S6D_PEM_GRAPH", "1", str),                    # hipGraph replay of the Net for small instance counts
    "pem_graph_max": ("S6D_PEM_GRAPH_MAX", 16, int),
    "pem_sampler": ("S6D_PEM_SAMPLER", "", str),                 # "library": torch sampling in the pre-processing
    "pem_pre": ("S6D_PEM_PRE", "", str),                         # "library": torch pre-processing
    "rpe_fold": ("S6D_RPE_FOLD", "1", str),
    "geo_presplit": ("S6D_GEO_PRESPLIT", "1", str),
    "desc_group": ("S6D_DESC_GROUP", "1", str),                  # descriptors of a frame group in one DINOv2 pass
    "gemm_res": ("S6D_GEMM_RES", "0", str),
    "lnfold": ("S6D_LNFOLD", "1", str),
    "qkv_layout": ("S6D_QKV_LAYOUT", "token", str),
    "disable_fused": ("S6D_DISABLE_FUSED", "", str),             # comma-separated kernel names the modules must not use (tests)
    "debug": ("S6D_DEBUG", "", str),
    "strict": ("S6D_STRICT", "0", str),                          # "1": a library branch on a CUDA tensor raises
}


@dataclasses.dataclass
class PrecisionPolicy:
    sam_dtype: str = "bf16"
    sam_gemm: str = "bf16"
    sam_decoder_dtype: str = "bf16"
    sam_mlp_rows: int = 0
    samdec_gemm: str = "kernel"
    samdec_t2i: str = "raw"
    amg_graph: str = "1"
    amg_profile: str = ""
    dino_dtype: str = "bf16"
    dino_gemm: str = "bf16"
    pem_vit_dtype: str = "fp32"
    pem_geo_dtype: str = "fp32"
    pem_f16_guard: str = "1"
    pem_graph: str = "1"
    pem_graph_max: int = 16
    pem_sampler: str = ""
    pem_pre: str = ""
    rpe_fold: str = "1"
    geo_presplit: str = "1"
    desc_group: str = "1"
    gemm_res: str = "0"
    lnfold: str = "1"
    qkv_layout: str = "token"
    disable_fused: str = ""
    debug: str = ""
    strict: str = "0"

    @classmethod
    def from_env(cls, environ=None):
        e = os.environ if environ is None else environ
        return cls(**{f: typ(e.get(var, default)) for f, (var, default, typ) in _ENV.items()})

    @classmethod
    def benched(cls, **kw):
        """What bench.py measures as BASELINE configs[1]: bf16 SAM ViT-H / mask decoder / DINOv2 (the defaults) and the PEM's ViT-B
        extractor in IEEE half (pose within 1e-3 / 1e-3 mm of the reference on the well-conditioned golden, tests/test_gpu_pem.py);
        everything else as the environment says."""
        p = cls.from_env()
        p.pem_vit_dtype = "fp16"
        for k, v in kw.items():
            setattr(p, k, v)
        return p


_current = PrecisionPolicy.from_env()
_version = 0          # bumped on every change: caches keyed on the policy (captured graphs, ops.have) compare it


def current():
    return _current


def version():
    return _version


def set_policy(p):
    global _current, _version
    assert isinstance(p, PrecisionPolicy)
    _current = p
    _version += 1


def set(**kw):
    """Change fields of the current policy in place (unknown names raise)."""
    global _version
    for k, v in kw.items():
        if k not in _ENV:
            raise AttributeError(f"PrecisionPolicy has no field {k!r}")
        setattr(_current, k, v)
    _version += 1


def reload():
    """Re-read the S6D_* variables (code that still configures through the environment: the test-suite's monkeypatch, tools/)."""
    set_policy(PrecisionPolicy.from_env())


@contextlib.contextmanager
def use(**kw):
    """with policy.use(sam_gemm="fp8", dino_gemm="fp8"): ...  -- the previous policy comes back afterwards."""
    global _current, _version
    old = _current
    _current = dataclasses.replace(old, **kw)
    _version += 1
    try:
        yield _current
    finally:
        _current = old
        _version += 1


# ---- observability of the library branches --------------------------------------------------------------------------------------
class StrictError(RuntimeError):
    pass


_hits = {}


def guard(site, **conditions):
    """The `if` in front of a kernel path: True when every condition holds.  `cuda=` (if given) says whether the operands live on the
    GPU: a failure there is a fall-back to the library on the device -- recorded, and an error under strict mode; a failure on CPU
    tensors (the host tier of the tests, the CPU baseline) is the reference path doing its job and is not recorded."""
    for name, ok in conditions.items():
        if not ok:
            if conditions.get("cuda", True) and name != "cuda":
                key = (site, name)
                _hits[key] = _hits.get(key, 0) + 1
                if _current.strict == "1":
                    raise StrictError(f"{site}: the library branch would run on the GPU because guard `{name}` failed "
                                      f"(conditions: {', '.join(f'{k}={bool(v)}' for k, v in conditions.items())})")
            return False
    return True


def library_branch_hits():
    """{(site, failed guard): count} of library branches taken on CUDA tensors since the last reset."""
    return dict(_hits)


def reset_library_branch_hits():
    _hits.clear()
