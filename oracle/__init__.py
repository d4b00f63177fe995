"""Oracle package — TEST INFRASTRUCTURE ONLY (never imported by yadcc_amd/).

* dispatch_oracle.{h,c}: plain-C restatement of yadcc's TaskDispatcher placement
  (reference yadcc/scheduler/task_dispatcher.cc:283-451).
* ref_driver.{h,cc} + shims/: the reference's own translation units compiled
  verbatim into oracle/_ref/libyadcc_ref.so.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use these.
"""
