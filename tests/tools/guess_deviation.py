"""Analysis tool (CPU model, no GPU): how far the level guesses of the matching passes are from
the start states the chunks really have. Usage: python tests/tools/guess_deviation.py [cfg3] [512]

On cfg3 (1M requests x 8k servants, chunks of 512): 1698 of 1954 chunks are off by 2-10 list
positions summed over the 30 classes (healed within the 16 warm-up requests of pass 0), and the
four chunks around the request that takes the last slot of the dedicated tier (rank 336935 =
chunk 658) are off by 58 / 218 / 298 / 126: there the true class states leave the level for
~2000 requests, which have to be replayed one after the other."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from tests.model import modelbind as M  # noqa: E402
from yadcc_amd import synth  # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
    chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    sv, tk = synth.make_config(cfg)
    k = -(-len(tk["env_id"]) // chunk)
    dev_sum = np.zeros(k, np.uint32)
    dev_max = np.zeros(k, np.uint32)
    M.lib().model_set_deviation_out(dev_sum.ctypes.data_as(C.c_void_p), dev_max.ctypes.data_as(C.c_void_p))
    _, _, _, st = M.dispatch(sv, tk, chunk)
    M.lib().model_set_deviation_out(None, None)
    print("%s: %d chunks of %d, %d classes, %d slots; %d rounds, %d chunk replays on the CPU model"
          % (cfg, st.n_chunks, chunk, st.n_classes, st.n_slots, st.rounds, st.chunk_sims))
    print("chunks whose level guess is off: %d" % int((dev_sum > 0).sum()))
    print("histogram of the summed deviation (0..19, 20+):", np.bincount(np.minimum(dev_sum, 20), minlength=21))
    for i in np.nonzero(dev_sum > 20)[0]:
        print("  chunk %d: sum %d, largest class %d" % (i, dev_sum[i], dev_max[i]))


if __name__ == "__main__":
    main()
