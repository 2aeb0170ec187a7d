# Round 5: the kernels of one step of the N > 1 headline path with one rank and the peer exchange (rocprofv3 --kernel-trace): what the +13 us over the plain step are made of
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r05x; mkdir -p $O
cd $GRAFT_REPO_ROOT
F="--steps 50 --warmup 5 --no-configs --no-c4 --no-traffic --no-big-source --no-cpu-baseline --no-cold --device-warmup-ms 20"
rm -rf /tmp/px; GP_BENCH_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29791 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/px -o t -- python bench.py $F --exchange peer > $O/trace.log 2>&1
python - <<'PY' | tee $O/trace_summary.txt
import csv, glob
f = sorted(glob.glob("/tmp/px/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
for r in rows:
    r["t0"], r["t1"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["t0"])
ex = [i for i, r in enumerate(rows) if "peer_exchange_kernel" in r["Kernel_Name"]]
print(len(rows), "dispatches,", len(ex), "exchange kernels; the last three steps:")
for i in ex[-3:]:
    j = i
    while j > 0 and "peer_exchange_kernel" not in rows[j - 1]["Kernel_Name"]:
        j -= 1
    t0 = rows[j]["t0"]
    for r in rows[j:i + 1]:
        print(f"  +{(r['t0'] - t0) / 1e3:6.1f} us  dur {(r['t1'] - r['t0']) / 1e3:6.1f}  q{r['Queue_Id']}  grid {r['Grid_Size_X']:>7}  {r['Kernel_Name'].split('(')[0].replace('void gp::', '')[:70]}")
    print("  step's kernels span", round((rows[i]["t1"] - t0) / 1e3, 1), "us")
PY
