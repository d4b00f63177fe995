"""yadcc_amd — MI355X-native task-dispatch path of Tencent/yadcc's scheduler.

Only what the hot path needs:
  csrc/          HIP kernels (gfx950) + the extern "C" boundary (include/yadcc_dispatch.h)
                 + the C++ host class mirroring the reference TaskDispatcher
  binding.py     ctypes view of libydc.so (fails loudly if the library or a GPU is missing):
                 batch dispatch, streaming ticks, the multi-GPU group
  dispatcher.py  ctypes view of the host class (ydc_td_*)
  pack.py        raw servant personalities -> C-ABI columns
  synth.py       seeded synthetic (pool, batch) snapshots of BASELINE.json's configs
  streaming.py   the synthetic event stream of the streaming configuration (bench + tests)
There is deliberately no CPU implementation in this package.
"""
__all__ = ["binding", "dispatcher", "pack", "streaming", "synth"]
