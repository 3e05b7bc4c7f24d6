#!/bin/bash
# Kernel trace of the video driver's loop at fhd (BASELINE.json configs[4], device PNG encoder) and of the encoder alone
# at fhd / 4k / 8k; run on the GPU box: tools/profile_video.sh  -> gpurun_out/prof_video_loop.md, prof_png.md
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof_video_loop
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/tools/exp_video.py 900 -1 > $OUT.log 2>&1 || exit 1
python3 $ROOT/tools/summarize_video_prof.py $OUT 900 "$(grep 'render_video:' $OUT.log)" > $ROOT/gpurun_out/prof_video_loop.md
cat $ROOT/gpurun_out/prof_video_loop.md
bash $ROOT/tools/exp_png_prof.sh > $ROOT/gpurun_out/prof_png.md 2>&1
cat $ROOT/gpurun_out/prof_png.md
