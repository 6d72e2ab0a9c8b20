set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final4; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export BNN_MI355X_LANES=1 BNN_MI355X_NO_WARMUP=1
for NET in lfcW1A1 lfcW1A2; do
  B="python3 $R/bench.py --network $NET --steps 5 --warmup 1 --no-cpu-baseline --no-extras"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_$NET -- $B > $O/pmc_fetch_$NET.json 2>$O/pmc_fetch_$NET.err
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_$NET -- $B > $O/pmc_write_$NET.json 2>$O/pmc_write_$NET.err
  python3 $R/tools/pmc_summary.py $O/pmc_fetch_$NET $O/pmc_write_$NET 131072 > $O/pmc_traffic_$NET.txt
  cat $O/pmc_traffic_$NET.txt
  python3 $R/tools/make_traffic_json.py $NET $O/pmc_traffic_$NET.txt "profiles/r04_pmc_traffic_$NET.txt: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (round 4, one compute lane, no load-time warm-up; the stages the bench's event pass runs, i.e. the staged form next to the one-launch kernel), FETCH_SIZE doubled (gfx950 correction, MI355X_MICROARCH.md HBM section)" $O/traffic.json
  cp $(ls $O/pmc_fetch_$NET/*/*counter_collection.csv | head -1) $O/pmc_fetch_counter_collection_$NET.csv
  cp $(ls $O/pmc_write_$NET/*/*counter_collection.csv | head -1) $O/pmc_write_counter_collection_$NET.csv
  rm -rf $O/pmc_fetch_$NET $O/pmc_write_$NET
done
echo partD done
