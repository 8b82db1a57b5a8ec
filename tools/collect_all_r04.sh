set -o pipefail
bash tools/collect_profiles.sh r04_cfg3 > gpurun_out/r04_cfg3.log 2>&1; echo "cfg3 rc=$?"
bash tools/collect_profiles.sh r04_cfg5 --config cfg5 > gpurun_out/r04_cfg5.log 2>&1; echo "cfg5 rc=$?"
