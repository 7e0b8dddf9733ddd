# Experiment: MIOpen exhaustive tuning (MIOPEN_FIND_ENFORCE=SEARCH_DB_UPDATE) of the YOLOv5s library convolutions into a copy of
# the shipped find-db, then the bench with that copy.  Writes gpurun_out/miopen_db_tuned/.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/miopen_db_tuned && cp heltondetection_amd/miopen_db/*.txt gpurun_out/miopen_db_tuned/
export MIOPEN_USER_DB_PATH=$GRAFT_REPO_ROOT/gpurun_out/miopen_db_tuned
( while true; do sleep 60; echo "[tune heartbeat] $(date +%T)"; done ) &
HB=$!
MIOPEN_FIND_ENFORCE=4 timeout -k 10 660 python bench.py --steps 5 --warmup 5 --no-graph --no-legs --no-micro --no-cpu-baseline --no-roofline > gpurun_out/tune1.log 2> gpurun_out/tune1.err
echo "tune rc=$?"
kill $HB
grep "warm-up step 0" gpurun_out/tune1.err | cut -c1-100
ls -la gpurun_out/miopen_db_tuned | head
python bench.py --steps 20 --warmup 5 --no-legs --no-micro --no-cpu-baseline --no-roofline 2> gpurun_out/tune2.err | grep -o "\"ms_per_step\": [0-9.]*"
