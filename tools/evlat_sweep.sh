# Developer tool (GPU box): syn_policy_eval_batch / syn_eval_ctx_* latency by batch size under the debug knobs of the transfer path
export SYN_DEBUG=1
mkdir -p gpurun_out
echo "pageable transfers + throughput kernel (round 3's path)"; SYN_EVAL_PAGEABLE=1 python3 tools/eval_latency.py
echo "default (results in place <= 4,096, completion polled <= 1,024)"; python3 tools/eval_latency.py
echo "results always by DMA"; SYN_EVAL_ZC_OUT=0 python3 tools/eval_latency.py
echo "results always in place"; SYN_EVAL_ZC_OUT=2000000 python3 tools/eval_latency.py
echo "completion never polled"; SYN_EVAL_POLL_MAX=0 python3 tools/eval_latency.py
echo "completion always polled"; SYN_EVAL_POLL_MAX=2000000 python3 tools/eval_latency.py
