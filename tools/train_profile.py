"""Developer tool (GPU): three training steps with SYN_TRAIN_PROFILE=1 set print the per-phase cycle stamps of the
training kernel. Usage: SYN_TRAIN_PROFILE=1 python tools/train_profile.py"""
import os
os.environ["SYN_DEBUG"] = "1"  # developer knobs (SYN_LANES, SYN_PROFILE, ...) are honoured only with SYN_DEBUG=1
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import synthesis_amd as sa
blob = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden', 'c4net_blob_f32.npy')); g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden', 'train_torch_goldens.npz'))
e = sa.Engine(64, 64); e.load_weights(blob); e.trainer_init(blob)
for s in range(3): e.train_step(g['my_bb'][s], g['op_bb'][s], g['target_pi'][s], g['target_v'][s], 1e-3)
