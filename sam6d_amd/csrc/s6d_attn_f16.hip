// IEEE-half build of the fused attention kernels (csrc/s6d_attn.hip compiled with S6D_ATTN_F16 = 1, namespace s6d_h): exports
// s6d_seq_attention_f16, the sequence attention of the PEM's ViT-B (Pose_Estimation_Model/model/feature_extraction.py:17-35 on
// timm's Attention) in half precision -- its 11-bit significand keeps the extractor's features within 1e-3 of the fp32 ones,
// which the 1e-3 mm translation bar of the matcher needs (bf16: 7.6e-3).
#define S6D_ATTN_F16 1
#include "s6d_attn.hip"
