"""One frame through all five (mini / seeded) models on the emulator: the body of tests/test_gpu_zz_pipeline.py with every
kernel of the chain running from its source on the host.  Minutes, not seconds: only with S6D_EMU_SLOW=1."""
import os

import pytest
import torch

from tests import test_gpu_zz_pipeline as T


@pytest.mark.skipif(os.environ.get("S6D_EMU_SLOW") != "1", reason="several minutes on the emulator: set S6D_EMU_SLOW=1")
def test_frame_through_all_five_models_on_the_emulator(emu, monkeypatch):
    # the bf16 paths rely on CUDA autocast to cast the weights per op; without CUDA the fp32 paths of the three ViTs are used
    # (their fused bf16 kernels have their own emulator tests), everything else is the device path
    for k in ("S6D_SAM_DTYPE", "S6D_DINO_DTYPE", "S6D_SAM_DECODER_DTYPE", "S6D_PEM_VIT_DTYPE"):
        monkeypatch.setenv(k, "fp32")
    T.run_frame(torch.device("cpu"), group_check=os.environ.get("S6D_EMU_GROUP") == "1")       # + S6D_EMU_GROUP=1: run_group too (~50 min)
