"""tests/test_gpu_zz_frame.py on the emulator: the ISM scoring kernels, the hand-off, the threshold and the PEM pre-processing
kernels of one frame in the reference's order against the reference-made golden (the Net itself: S6D_EMU_SLOW=1)."""
import os

import torch

from tests import test_gpu_zz_frame as T


def test_frame_chain_on_the_emulator(emu):
    T.run_chain(torch.device("cpu"), net_check=os.environ.get("S6D_EMU_SLOW") == "1")
