#!/bin/bash
# The -m gpu suite under the knobs that move work between code paths (GPU box).  One line per setting.
# usage: tools/knob_matrix.sh <outdir>
export MIPT_DEBUG=1
OUT=${1:-gpurun_out/knobs}; mkdir -p "$OUT"
run() {
  tag=$1; shift
  env "$@" timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 240 --timeout-method thread \
    -k "not bench and not ray_cli and not ten_million and not hand_rays_back and not two_launch_streams and not mesh_scene_kernels and not analytic_primitives_beside" \
    > "$OUT/$tag.log" 2>&1
  rc=$?
  echo "$tag: rc=$rc $(tail -1 "$OUT/$tag.log")"
  return $rc
}
run lds2 MIPT_LDS_STACK=2 &&
run lds3 MIPT_LDS_STACK=3 &&
run builder_gpu MIPT_BUILDER=gpu &&
run overlap_all MIPT_OVERLAP_RAYS=1073741824 &&
run overlap_all_fuse_all MIPT_OVERLAP_RAYS=1073741824 MIPT_FUSE_RAYS=16777216 &&
run fuse_off MIPT_FUSE_RAYS=0 &&
run fuse_budget2 MIPT_FUSE_RAYS=16777216 MIPT_FUSE_BUDGET=2 &&
run tail_everywhere MIPT_TAIL_RAYS=1000000 &&
run plain_loop MIPT_OVERLAP=0 MIPT_TAIL_RAYS=0 &&
run vis_bytes_overlap MIPT_VIS_BYTES=1 MIPT_OVERLAP_RAYS=1073741824 &&
run fat_any MIPT_LEAN_ANY=0 MIPT_VIS_BYTES=0
