set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for t in 8192 16384; do
for tune in "j5_dm_fused=1" "j5_dm_fused=0"; do
  args=""; for kv in $tune; do args="$args --tune $kv"; done
  timeout -k 10 200 python scripts/dm_overhead.py --tile $t $args --out gpurun_out/dm_ab.json 2>&1 | grep -E '"tuning"|"plain"|"serial"|"overlapped"|"pipelined"|_efficiency' | tr -d '\n'; echo
done; done
