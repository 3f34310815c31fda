set -e
for th in 4 8 12 16; do BRBD_HOST_THREADS=$th timeout -k 10 60 python tools/host_path_bench.py; done
for ch in 4096 8192 32768; do BRBD_HOST_CHUNK=$ch timeout -k 10 60 python tools/host_path_bench.py; done
BRBD_HOST_THREADS=16 BRBD_HOST_CHUNK=8192 timeout -k 10 60 python tools/host_path_bench.py
HOST_B=1048576 timeout -k 10 60 python tools/host_path_bench.py
HOST_B=1048576 BRBD_HOST_THREADS=16 timeout -k 10 60 python tools/host_path_bench.py
HOST_B=1048576 BRBD_HOST_THREADS=0 timeout -k 10 60 python tools/host_path_bench.py
