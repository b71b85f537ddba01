"""sam6d_amd -- MI355X-native per-frame inference hot path of SAM-6D.

Package layout (only what the hot path needs):
  csrc/        hand-written gfx950 HIP kernels + the C ABI (include/sam6d_hip.h)
  ops.py       torch-tensor front-end of the C ABI
  pointnet2/   drop-in for the reference's ``pointnet2._ext`` pybind module
  pem/         drop-in Pose_Estimation_Model modules (Net, ViTEncoder, matching heads)
  sam/         drop-in SAM ImageEncoderViT
  ism/         drop-in ISM scoring classes
  utils/       seeded weights / synthetic frames, ADD(-S) metrics, frame sharding
"""
__version__ = "0.1.0"
