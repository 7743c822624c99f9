mkdir -p gpurun_out/r06g
B=$PWD/deeppointmap_amd/csrc/build
DPM_LIB=$B/libdpm_nt3.so timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -k fps 2>&1 | tail -2
for v in nt1 nt3; do for b in 64 1; do DPM_LIB=$B/libdpm_$v.so python scripts/fps_algo_bench.py $b 5 2>&1 | tail -1; done; done
python scripts/fps_algo_bench.py 64 5 | tail -1
bash scripts/debug/lib_ab.sh gpurun_out/r06g/nt 2 60 shipped=shipped nt1=$B/libdpm_nt1.so nt3=$B/libdpm_nt3.so
