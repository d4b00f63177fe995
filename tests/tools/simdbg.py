"""Developer aid: prints the matching passes' outcome per batch (YDC_DEBUG_SIM=1: which passes
changed an end state, number of chunk replays)."""
import os, sys
sys.path.insert(0, os.getcwd())
os.environ["YDC_DEBUG_SIM"] = "1"
from yadcc_amd import binding, pack, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
sv, tk = synth.make_config(cfg)
ctx = binding.Context()
ctx.upload_servants(pack.to_abi_columns(sv))
for _ in range(3):
    ctx.dispatch(tk, want_util=False, want_running=False)
print(ctx.stats())
