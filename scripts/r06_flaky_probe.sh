mkdir -p gpurun_out/r06flaky
for i in 1 2 3 4 5 6 7 8; do
  MASTER_ADDR=127.0.0.1 MASTER_PORT=2954$i RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 python bench.py --gpus 1 --force-dist --no-afr --steps 6 --warmup 2 --spinup-ms 50 --no-cpu-baseline > gpurun_out/r06flaky/out_$i.json 2> gpurun_out/r06flaky/err_$i.txt
  rc=$?
  echo "run $i rc=$rc"
  if [ $rc -ne 0 ]; then grep -v "frame #" gpurun_out/r06flaky/err_$i.txt | head -40; fi
done
