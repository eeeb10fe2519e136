cd /root/repo; export TMPDIR=/tmp GFHIP_EXPERIMENTS=1; O=gpurun_out/r05_h_knobs; mkdir -p $O
for s in 5 2 10 5; do
  echo "== spmm_slack=$s"; timeout 60 python tools/hop_probe.py cfg4 5 spmm_slack=$s v:spmm_pfd=16 v:spmm_pfd=12 v:spmm_pfd=20 v:spmm_pfd=24 v:spmm_pfd=16 2>&1 | grep -v amdgpu.ids
done | tee $O/slack_pfd_final_image.log
