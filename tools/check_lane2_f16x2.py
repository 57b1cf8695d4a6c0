import os, sys
os.environ["SYN_DEBUG"]="1"; os.environ["SYN_LANES2"]="8"
sys.path.insert(0,".")
import numpy as np, synthesis_amd as sa
from tests import oracle_lib
from tests.oracle_lib import parity_rollout_config
from tests.test_gpu_parity import assert_selfplay_equal
orc=oracle_lib.load()
w=np.load("tests/golden/c4net_trained_f32.npy")
eng=sa.Engine(concurrent_games=1024, max_explores=200); eng.load_weights(w); eng.set_network_arithmetic("f16x2")
got=eng.selfplay(sa.parity_rollout_config(100), base_seed=9, n_games=1500)
print("shape",eng.last_launch_shape())
ref=orc.c4_selfplay(parity_rollout_config(100), w, 9, 300, threads=8, nn_mode=orc.ACC_F16X2)
assert_selfplay_equal({k:got[k][:300] for k in ("plies","states_bb","pis","vs","actions","root_nodes","final_kind")}, ref, "lane2 f16x2")
print("lane2 f16x2 parity ok")
