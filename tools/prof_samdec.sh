cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
cat > /tmp/dec_once.py <<'PY'
import os, sys, torch
sys.path.insert(0, ".")
os.environ["S6D_SAM_DECODER_DTYPE"] = "bf16"
from sam6d_amd.utils import seeded, synth
from sam6d_amd.sam.mask_decoder import build_sam_decoder
cfg = dict(dim=256, emb=64, img=1024)
m = seeded.load_seeded(build_sam_decoder(), 1).cuda()
inp = {k: v.cuda() for k, v in synth.sam_decoder_inputs(cfg, 1024, 3).items()}
def frame():
    with torch.no_grad():
        for a in range(0, 1024, 256):
            s, d = m.prompt_encoder(points=(inp["points"][a:a+256], inp["labels"][a:a+256]), boxes=None, masks=None)
            m.mask_decoder(image_embeddings=inp["emb"], image_pe=m.prompt_encoder.get_dense_pe(), sparse_prompt_embeddings=s, dense_prompt_embeddings=d, multimask_output=True)
for _ in range(4): frame()
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dec -o dec -- python /tmp/dec_once.py > /dev/null 2>&1
mkdir -p gpurun_out/prof; cp $(find /tmp/prof_dec -name "*kernel_stats.csv" | head -1) gpurun_out/prof/samdec_v2_kernel_stats.csv
