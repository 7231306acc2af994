TAG=r15
ulimit -c 0
mkdir -p gpurun_out
bash scripts/profile.sh $TAG > gpurun_out/${TAG}_profile.log 2>&1
python3 bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python3 scripts/traffic.py gpurun_out/prof_$TAG gpurun_out/${TAG}_bench.json gpurun_out/${TAG}_traffic.json > gpurun_out/${TAG}_traffic.log 2>&1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver.json 2> gpurun_out/${TAG}_bench_driver.err
tail -c 300 gpurun_out/${TAG}_bench_driver.json
