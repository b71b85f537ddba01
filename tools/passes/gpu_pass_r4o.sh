# round 4: kernel statistics of the whole-frame demo at HEAD (decoder with raw attention, sequence attention with two workgroups per CU)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
rm -rf /tmp/fd && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fd -o fd -- python tools/frame_demo.py > gpurun_out/prof/r04_frame_demo.txt 2>&1
cp $(find /tmp/fd -name "*kernel_stats.csv" | head -1) gpurun_out/prof/r04_frame_demo_kernel_stats.csv
tail -1 gpurun_out/prof/r04_frame_demo.txt | cut -c1-600
