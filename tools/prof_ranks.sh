#!/bin/bash
# tools/prof_ranks.sh <tag> <N> [bench args]: the N-rank bench (one process per GPU, RCCL gather) with EVERY rank under rocprofv3:
#   pass 1: --kernel-trace --stats per rank      -> gpurun_out/<tag>/rank<r>_kernel_stats.csv (+ the JSON line of rank 0: bench_gpus<N>.json)
#   pass 2, 3: --pmc FETCH_SIZE / --pmc WRITE_SIZE per rank (counter passes carry --kernel-trace only, as the pool requires)
#              -> rank<r>_pmc_fetch.csv / rank<r>_pmc_write.csv, and hbm_gbs_per_rank.txt: HBM GB/s of the traversal (rp_k_extend + rp_k_connect)
#              and shading (rp_k_shade) kernels per rank = (2 x FETCH_SIZE + WRITE_SIZE) KiB / kernel time -- the "achieved HBM GB/s at
#              1/2/4/8 GPUs" leg of north_star. On a one-GPU box: tools/prof_ranks.sh <tag> 2 --same-device (gloo fall-back, same scripts).
set -u
TAG=${1:-r03}; N=${2:-2}; shift 2 || true
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
PORT=$((29600 + RANDOM % 200))
run_pass() { # <name> <rocprofv3 options...> -- one torch.distributed.run whose ranks are each wrapped in rocprofv3
  local NAME=$1; shift
  rm -rf /tmp/ranks_$NAME
  cat > /tmp/rank_wrap_$NAME.sh <<W
#!/bin/bash
exec rocprofv3 $@ -d /tmp/ranks_$NAME/rank\$RANK -o r\$RANK --output-format csv -- python $R/bench.py --gpus $N --steps 40 --warmup 4 --no-cpu-baseline ${BENCH_ARGS:-}
W
  chmod +x /tmp/rank_wrap_$NAME.sh
  timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT --no-python /tmp/rank_wrap_$NAME.sh > $O/ranks_$NAME.log 2>&1
  PORT=$((PORT + 3))
}
BENCH_ARGS="$*"
run_pass stats --kernel-trace --stats
grep '^{' $O/ranks_stats.log | tail -1 > $O/bench_gpus$N.json
for r in $(seq 0 $((N - 1))); do cp $(find /tmp/ranks_stats/rank$r -name "*kernel_stats.csv" | head -1) $O/rank${r}_kernel_stats.csv 2>/dev/null; done
run_pass fetch --kernel-trace --pmc FETCH_SIZE
run_pass write --kernel-trace --pmc WRITE_SIZE
python3 - $O $N <<'PY'
import csv, glob, sys, collections
out, n = sys.argv[1], int(sys.argv[2])
lines = []
for r in range(n):
    kib = collections.defaultdict(float); ns = collections.defaultdict(float)
    for name, scale in (("fetch", 2.0), ("write", 1.0)):   # gfx950: FETCH_SIZE reads 1/2 of the bytes (MI355X_MICROARCH.md)
        for fn in glob.glob("/tmp/ranks_%s/rank%d/**/*counter_collection.csv" % (name, r), recursive=True):
            seen = set()
            for row in csv.DictReader(open(fn)):
                k = row["Kernel_Name"]
                fam = "traversal" if ("rp_k_extend" in k or "rp_k_connect" in k) else ("shading" if "rp_k_shade" in k else None)
                if fam is None:
                    continue
                kib[fam] += scale * float(row["Counter_Value"])
                if name == "fetch" and row["Dispatch_Id"] not in seen and row.get("End_Timestamp") and row.get("Start_Timestamp"):
                    seen.add(row["Dispatch_Id"]); ns[fam] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
        if name == "fetch" and not ns:   # older CSV layouts keep the time stamps in the kernel trace of the same pass
            for fn in glob.glob("/tmp/ranks_fetch/rank%d/**/*kernel_trace.csv" % r, recursive=True):
                for row in csv.DictReader(open(fn)):
                    k = row["Kernel_Name"]
                    fam = "traversal" if ("rp_k_extend" in k or "rp_k_connect" in k) else ("shading" if "rp_k_shade" in k else None)
                    if fam:
                        ns[fam] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
    for fam in ("traversal", "shading"):
        if ns[fam] > 0:
            lines.append("rank %d %-9s HBM %.1f GB/s (%.1f MB in %.2f ms of kernel time, counter pass)" % (r, fam, kib[fam] * 1024 / ns[fam], kib[fam] * 1024 / 1e6, ns[fam] / 1e6))
open(out + "/hbm_gbs_per_rank.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
