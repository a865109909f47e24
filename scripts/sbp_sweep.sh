# Dev tool (GPU box): admm_hip_parbp at the C5 shape under different workgroup counts / list shares.
for cfg in "512 4" "512 1" "512 8" "512 16" "256 4" "1024 4"; do set -- $cfg; echo "WGS=$1 SHARE=$2: $(ADMM_HIP_SBP_WGS=$1 ADMM_HIP_SBP_SHARE=$2 python scripts/bench_configs.py c5parbp 2>&1 | tail -1 | cut -c87-200)"; done
