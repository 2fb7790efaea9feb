out=gpurun_out/sk1; mkdir -p $out
python -m pytest tests/test_gpu_ops.py -q -m gpu -k "splitk or conv2d_vs_fp64 or eight_wave or modulated" -x > $out/tests.txt 2>&1; tail -5 $out/tests.txt
python tools/bench_small.py > $out/small_fused.txt 2>&1
RICK_TUNE=2=0 python tools/bench_small.py > $out/small_unfused.txt 2>&1
paste -d'\n' $out/small_fused.txt $out/small_unfused.txt | grep -v Warn | cut -c1-200
