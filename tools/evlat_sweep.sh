export SYN_DEBUG=1
mkdir -p gpurun_out
echo "pageable"; SYN_EVAL_PAGEABLE=1 python3 tools/eval_latency.py
echo "default (in 16384 / out 1024)"; python3 tools/eval_latency.py
echo "all DMA"; SYN_EVAL_ZC_IN=0 SYN_EVAL_ZC_OUT=0 python3 tools/eval_latency.py
echo "all in place"; SYN_EVAL_ZC_IN=2000000 SYN_EVAL_ZC_OUT=2000000 python3 tools/eval_latency.py
echo "in place in, DMA out"; SYN_EVAL_ZC_IN=2000000 SYN_EVAL_ZC_OUT=0 python3 tools/eval_latency.py
