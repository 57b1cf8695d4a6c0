"""Writes tests/golden/c4net_blob.ot: the committed fixed-seed Connect4Net blob (tests/golden/c4net_blob_f32.npy) as the VarStore archive
libtorch's torch::serialize::OutputArchive produces — the call tch's `VarStore::save` makes (`at_save_multi`: one `archive.write(name,
tensor, /*is_buffer=*/false)` per variable, then `archive.save_to(path)`; synthesis/src/alpha_zero.rs:37,97 `vs.save(...)`), with the
variable names of study-connect4/src/policies.rs:20-24. Built with the torch of the authoring image (a small C++ extension: the Python API
has no handle on OutputArchive); nothing of this script runs in the tests. No `.ot` written by the reference exists here (no Rust
toolchain), so the reader (synthesis_amd/weights.py::load_ot) is pinned to libtorch's writer, not to a reference-produced file.
"""
import os

import numpy as np
import torch
from torch.utils.cpp_extension import load_inline

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = r'''
#include <torch/torch.h>
#include <torch/serialize/archive.h>
void save_ot(std::vector<std::string> names, std::vector<torch::Tensor> tensors, std::string path) {
    torch::serialize::OutputArchive archive;
    for (size_t i = 0; i < names.size(); i++) archive.write(names[i], tensors[i], /*is_buffer=*/false);
    archive.save_to(path);
}
'''
DIMS = [63, 128, 96, 64, 48, 12]
ext = load_inline("ot_writer", cpp_sources=SRC, functions=["save_ot"], verbose=False)
blob = np.load(os.path.join(HERE, "c4net_blob_f32.npy"))
names, tensors, off = [], [], 0
for l in range(5):
    w = blob[off:off + DIMS[l] * DIMS[l + 1]].reshape(DIMS[l + 1], DIMS[l]); off += w.size
    b = blob[off:off + DIMS[l + 1]]; off += b.size
    names += [f"l_{l + 1}.weight", f"l_{l + 1}.bias"]
    tensors += [torch.from_numpy(w.copy()), torch.from_numpy(b.copy())]
ext.save_ot(names, tensors, os.path.join(HERE, "c4net_blob.ot"))
print("written", os.path.getsize(os.path.join(HERE, "c4net_blob.ot")), "bytes")
# The names as tch's VarStore::save hands them to that call: Tensor::save_multi replaces every '.' by '|' before at_save_multi
# (a TorchScript attribute name cannot hold a dot; Tensor::load_multi maps it back) — the on-disk form of a reference `model_i.ot`.
ext.save_ot([n.replace(".", "|") for n in names], tensors, os.path.join(HERE, "c4net_blob_tch_names.ot"))
print("written", os.path.getsize(os.path.join(HERE, "c4net_blob_tch_names.ot")), "bytes")
