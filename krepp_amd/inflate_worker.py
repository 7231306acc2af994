"""python -m krepp_amd.inflate_worker INDEX_DIR INDEX_GB DEVICE SEED OUT_DIR

Child process of krepp_amd.synth.inflate_in_child: inflates the table of the index at INDEX_DIR to INDEX_GB on the GPU
(synth.inflate_and_upload: same generator, same seeds as every earlier round) and writes inc.npy / cmer.npy (on-disk layout:
cumulative bucket ends, {enc32, se} pairs) to OUT_DIR.  Exists so that the benchmark process itself does not churn device memory
before it allocates its streams."""
import os
import sys

import numpy as np


def main():
    index_dir, index_gb, device, seed, out_dir = sys.argv[1], float(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    import torch

    from krepp_amd import capi, synth

    capi.load()
    torch.cuda.set_device(device)
    hx = capi.HostIndex(index_dir)
    dx, (inc, cmer) = synth.inflate_and_upload(torch, capi, hx, torch.device("cuda", device), device, index_gb, seed=seed)
    dx.close()
    np.save(os.path.join(out_dir, "inc.npy"), inc)
    np.save(os.path.join(out_dir, "cmer.npy"), cmer)
    hx.close()


if __name__ == "__main__":
    main()
