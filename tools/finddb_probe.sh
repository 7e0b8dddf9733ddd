export MIOPEN_USER_DB_PATH=$GRAFT_REPO_ROOT/gpurun_out/miopen_db
mkdir -p $MIOPEN_USER_DB_PATH
cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 5 --no-micro --no-cpu-baseline --no-roofline > gpurun_out/fdb_yolo1.log 2> gpurun_out/fdb_yolo1.err
grep -o "\"ms_per_step\": [0-9.]*" gpurun_out/fdb_yolo1.log | head -3
timeout -k 10 700 python bench.py --model frcnn --steps 10 --warmup 6 --no-micro --no-cpu-baseline --no-roofline > gpurun_out/fdb_frcnn1.log 2> gpurun_out/fdb_frcnn1.err
grep -o "\"ms_per_step\": [0-9.]*" gpurun_out/fdb_frcnn1.log | head -1
ls -la $MIOPEN_USER_DB_PATH | head; du -sh $MIOPEN_USER_DB_PATH
# second pass: the db is there, no search
python bench.py --steps 20 --warmup 5 --no-legs --no-micro --no-cpu-baseline --no-roofline 2> gpurun_out/fdb_yolo2.err | grep -o "\"ms_per_step\": [0-9.]*"
grep "warm-up step 0" gpurun_out/fdb_yolo1.err gpurun_out/fdb_yolo2.err | cut -c1-120
timeout -k 10 400 python bench.py --model frcnn --steps 10 --warmup 6 --no-micro --no-cpu-baseline --no-roofline 2> gpurun_out/fdb_frcnn2.err | grep -o "\"ms_per_step\": [0-9.]*"
grep "warm-up step 0" gpurun_out/fdb_frcnn1.err gpurun_out/fdb_frcnn2.err | cut -c1-120
