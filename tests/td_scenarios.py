"""Scenarios for the host class (the reference's TaskDispatcher surface), shared by
tests/test_task_dispatcher_gpu.py (libydc.so on the MI355X, -m gpu) and
tests/test_task_dispatcher_stub.py (the same C++ host class linked against the CPU stand-in
of the device API, tests/native — no GPU). `make(**kw)` builds a dispatcher.

(1) The reference's own gtest cases (yadcc/scheduler/task_dispatcher_test.cc:29-144 `All`,
    :146-186 `PreferDedicated`, :216-298 `LoadBalanceCase`) re-expressed call for call, real
    sleeps replaced by the injected clock + OnExpirationTimer.
(2) A seeded random event stream (heartbeats, grant batches, frees, lease renewals, servant
    reports, clock + timer, servant expiry) applied to our class and to the reference's
    translation units compiled verbatim (oracle/_ref); every observable answer must be
    identical. n_digests > 64 exercises the multi-word environment masks (the reference has
    no limit on distinct digests, task_dispatcher.h:93-94)."""
import numpy as np

from oracle import refbind as R
from yadcc_amd import dispatcher as D

G50 = 50 << 30


def golden_all(td):
    """task_dispatcher_test.cc:29-144."""
    td.keep_servant_alive("127.0.0.1:1234", ["digest"], 10, 10, 0, memory_available=G50)
    st, _, _ = td.wait_for_starting_new_task("127.0.0.1", "not found", timeout_in_ms=1000)
    assert st == D.ENV_NOT_FOUND
    tasks = []
    for _ in range(10):
        st, tid, loc = td.wait_for_starting_new_task("127.0.0.1", "digest", expires_in_ms=5000,
                                                     timeout_in_ms=1000)
        assert st == D.GRANTED and loc == "127.0.0.1:1234"
        tasks.append(tid)
    assert tasks == list(range(10))  # next_task_id starts at 0 (task_dispatcher.h:218)
    st, _, _ = td.wait_for_starting_new_task("127.0.0.1", "digest", timeout_in_ms=0)
    assert st == D.TIMEOUT
    assert not td.keep_task_alive(12345678, 1000)
    for t in tasks:
        assert td.keep_task_alive(t, 1000)
    assert td.notify_servant_running_tasks("127.0.0.1:1234", [tasks[0], 1000002, 1000003]) == [
        1000002, 1000003]
    td.clock_advance_ms(2000)
    td.on_expiration_timer()  # leases (1 s) expired -> zombies
    for t in tasks:
        assert not td.keep_task_alive(t, 1000)
    assert td.notify_servant_running_tasks("127.0.0.1:1234", tasks) == tasks
    td.keep_servant_alive("127.0.0.1:1234", ["digest"], 10, 10, 0, memory_available=G50,
                          expires_in_ms=1000)
    td.clock_advance_ms(2000)
    td.on_expiration_timer()  # servant expired
    st, _, _ = td.wait_for_starting_new_task("127.0.0.1", "digest", timeout_in_ms=0)
    assert st != D.GRANTED


def golden_prefer_dedicated(td):
    """task_dispatcher_test.cc:146-186."""
    td.keep_servant_alive("127.0.0.1:1234", ["digest"], 10, 10, 0, priority=D.PRIORITY_USER,
                          memory_available=G50)
    st, tid, loc = td.wait_for_starting_new_task("127.0.0.1", "digest")
    assert (st, loc) == (D.GRANTED, "127.0.0.1:1234")  # self allowed when alone
    td.free_task(tid)
    td.keep_servant_alive("192.168.0.1:1234", ["digest"], 10, 10, 2,
                          priority=D.PRIORITY_DEDICATED, memory_available=G50)
    st, tid, loc = td.wait_for_starting_new_task("127.0.0.1", "digest")
    assert (st, loc) == (D.GRANTED, "192.168.0.1:1234")


def golden_load_balance(td):
    """task_dispatcher_test.cc:216-298: expected picks 1,2,3,2,1,2,3."""
    from tests.test_oracle_golden import LB, LB_EXPECT
    loads = {}
    loc0, mt, npz, ld = LB[0]
    td.keep_servant_alive(loc0, ["Load Balance"], mt, npz, ld, memory_available=G50)
    st, _, _ = td.wait_for_starting_new_task("127.0.0.3", "Load Balance", timeout_in_ms=0)
    assert st == D.TIMEOUT  # overloaded servant: Timeout, not EnvironmentNotFound (:217-228)
    for loc, mt, npz, ld in LB[1:]:
        td.keep_servant_alive(loc, ["Load Balance"], mt, npz, ld, memory_available=G50)
        loads[loc] = [mt, npz, ld]
    got = []
    for _ in LB_EXPECT:
        st, _, loc = td.wait_for_starting_new_task("127.0.0.3", "Load Balance")
        assert st == D.GRANTED
        got.append([x[0] for x in LB].index(loc))
        loads[loc][2] += 1
        td.keep_servant_alive(loc, ["Load Balance"], *loads[loc], memory_available=G50)
    assert got == LB_EXPECT


def blocking_wait_is_woken_by_free_task(make):
    """A request that finds no free servant waits until FreeTask's notify_all
    (task_dispatcher.cc:116-118,187) — real clock, second thread."""
    import threading
    import time
    td = make(fake_clock=False)
    td.keep_servant_alive("10.0.0.1:1", ["d"], 1, 8, 0, memory_available=G50)
    st, tid, _ = td.wait_for_starting_new_task("9.9.9.9", "d", expires_in_ms=60000)
    assert st == D.GRANTED
    t0 = time.time()
    st2, _, _ = td.wait_for_starting_new_task("9.9.9.9", "d", timeout_in_ms=200)
    assert st2 == D.TIMEOUT and time.time() - t0 >= 0.19
    out = {}

    def waiter():
        out["r"] = td.wait_for_starting_new_task("9.9.9.9", "d", timeout_in_ms=5000)

    th = threading.Thread(target=waiter)
    th.start()
    time.sleep(0.2)
    td.free_task(tid)
    th.join(5)
    assert out["r"][0] == D.GRANTED and out["r"][2] == "10.0.0.1:1"
    td.close()


def heartbeat_wakes_nobody(make):
    """Reference wake-up order (task_dispatcher.cc:116-118,187,190-220): a heartbeat that adds
    capacity does not wake a parked waiter — the next caller takes the new capacity — and
    FreeTask does; a waiter whose request another thread completed returns at once."""
    import threading
    import time
    td = make(fake_clock=False)
    td.keep_servant_alive("10.0.0.1:1", ["d"], 1, 8, 0, memory_available=G50)
    st, tid, _ = td.wait_for_starting_new_task("9.9.9.9", "d", expires_in_ms=60000)
    assert st == D.GRANTED
    out = {}

    def waiter():
        out["r"] = td.wait_for_starting_new_task("9.9.9.9", "d", timeout_in_ms=20000)
        out["at"] = time.time()

    th = threading.Thread(target=waiter)
    th.start()
    time.sleep(0.2)
    td.keep_servant_alive("10.0.0.2:1", ["d"], 1, 8, 0, memory_available=G50)
    time.sleep(0.1)
    assert "r" not in out  # still parked
    st2, _, loc2 = td.wait_for_starting_new_task("9.9.9.9", "d", expires_in_ms=60000)
    assert (st2, loc2) == (D.GRANTED, "10.0.0.2:1")  # the new caller got the new servant
    time.sleep(0.1)
    assert "r" not in out
    t0 = time.time()
    td.free_task(tid)
    th.join(10)
    assert out["r"][0] == D.GRANTED and out["r"][2] == "10.0.0.1:1"
    assert out["at"] - t0 < 2.0  # woken by the free, not by its 20 s deadline
    td.close()


def _join_or_release(td, th, seconds=10):
    """Joins a waiter thread; one that is still parked is released through its (fake-clock)
    deadline, so a regression fails the assertion that follows instead of hanging the suite."""
    th.join(seconds)
    if th.is_alive():
        td.clock_advance_ms(4 * 3600000)
        th.join(seconds)


def timer_tick_wakes_parked_waiters(make):
    """OnExpirationTimer ends in UnsafeSweepOrphans -> UnsafeFreeTasks(sweeping) on EVERY tick
    (task_dispatcher.cc:478-496,518-520), and UnsafeFreeTasks notifies all waiters even for an
    empty list (:187): a parked waiter re-runs its loop at least once a second. So it is
    granted a servant that registered meanwhile, and gets EnvironmentNotFound once the last
    eligible servant has expired — without any FreeTask."""
    import threading
    import time
    td = make(fake_clock=True)
    td.keep_servant_alive("10.0.0.1:1", ["d"], 1, 8, 0, memory_available=G50, expires_in_ms=1000)
    st, tid, _ = td.wait_for_starting_new_task("9.9.9.9", "d", expires_in_ms=600000)
    assert st == D.GRANTED
    out = {}

    def waiter(key, digest):
        out[key] = td.wait_for_starting_new_task("9.9.9.9", digest, timeout_in_ms=3600000)

    # (1) parked, then a second servant registers: the heartbeat alone wakes nobody, the tick does
    th = threading.Thread(target=waiter, args=("a", "d"))
    th.start()
    time.sleep(0.2)
    td.keep_servant_alive("10.0.0.2:1", ["d"], 1, 8, 0, memory_available=G50, expires_in_ms=600000)
    time.sleep(0.15)
    assert "a" not in out
    td.on_expiration_timer()
    _join_or_release(td, th)
    assert out["a"][0] == D.GRANTED and out["a"][2] == "10.0.0.2:1"
    # (2) parked on a full pool; both servants expire: the tick turns the wait into
    # EnvironmentNotFound (:105-108) instead of leaving it asleep until its deadline
    th = threading.Thread(target=waiter, args=("b", "d"))
    th.start()
    time.sleep(0.2)
    assert "b" not in out
    td.clock_advance_ms(700000)
    td.on_expiration_timer()
    _join_or_release(td, th)
    assert out["b"][0] == D.ENV_NOT_FOUND
    td.close()


def concurrent_callers_are_combined(make):
    """Many threads calling WaitForStartingNewTask at once: every grant is distinct, the pool
    fills exactly, the rest time out — whatever the interleaving."""
    import threading
    td = make(fake_clock=False)
    for i in range(8):
        td.keep_servant_alive("10.0.0.%d:1" % i, ["d"], 5, 16, 0, memory_available=G50)
    res = []
    lock = threading.Lock()

    def worker(k):
        for _ in range(10):
            r = td.wait_for_starting_new_task("9.9.9.%d" % k, "d", expires_in_ms=60000)
            with lock:
                res.append(r)

    ths = [threading.Thread(target=worker, args=(k,)) for k in range(8)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    granted = [r for r in res if r[0] == D.GRANTED]
    assert len(granted) == 40 and len(res) == 80
    assert sorted(r[1] for r in granted) == list(range(40))
    per = {}
    for r in granted:
        per[r[2]] = per.get(r[2], 0) + 1
    assert all(v == 5 for v in per.values()) and len(per) == 8
    td.close()


def location_that_does_not_fit_is_an_error(make):
    """A grant whose servant location does not fit the caller's buffer is given back
    (YDC_ERR_CAPACITY), not handed out truncated."""
    from yadcc_amd import binding
    td = make()
    long_loc = "10.0.0.1:" + "9" * 200
    td.keep_servant_alive(long_loc, ["d"], 1, 8, 0, memory_available=G50)
    try:
        td.wait_for_starting_new_task("9.9.9.9", "d")
        raise AssertionError("expected an error")
    except binding.YdcError:
        pass
    assert td.dump_internals()["running_tasks"] == 0  # the slot is free again
    td.LOC = 512
    st, _, loc = td.wait_for_starting_new_task("9.9.9.9", "d")
    assert (st, loc) == (D.GRANTED, long_loc)
    td.close()


def lease_table_churn(make):
    """Tens of thousands of leases through the table that is indexed by the grant id (pages of
    4096 ids, freed and reused as their leases go): ids stay sequential (task_dispatcher.cc:127),
    survivors of freed pages stay renewable, freed ids are forgotten, long addresses and digests
    come back out of the dump, zombies are swept per servant, expiry frees the orphans."""
    td = make()
    digest = "d" * 64
    long_ip = "2001:0db8:85a3:0000:0000:8a2e:0370:7334"  # does not fit the record: pooled
    for i in range(3):
        td.keep_servant_alive("10.0.0.%d:8335" % (i + 1), [digest], 20000, 20000, 0,
                              memory_available=G50, expires_in_ms=10_000_000)
    ids = []
    for b in range(15):  # 15000 grants: pages 0..3
        st, got, locs = td.wait_for_starting_new_tasks(
            [long_ip if k == 0 else "172.16.%d.%d" % (b, k % 250) for k in range(1000)], [digest] * 1000,
            [0] * 1000, expires_in_ms=(50 if b == 3 else 10_000_000))
        assert (st == D.GRANTED).all()
        ids.extend(int(x) for x in got)
    assert ids == list(range(15000))
    keep = set(ids[::1000]) | set(range(3000, 4000))  # a few per page + the short leases of batch 3
    td.free_tasks([i for i in ids if i not in keep])
    dump = td.dump_internals()
    assert dump["running_tasks"] == len(keep) and len(dump["tasks"]) == len(keep)
    assert dump["tasks"]["7000"]["requestor_ip"] == long_ip and dump["tasks"]["7000"]["compiler_digest"] == digest
    assert dump["tasks"]["3001"]["requestor_ip"] == "172.16.3.1"
    assert all(td.keep_task_alive(i, 10_000_000) for i in ids[::1000])
    assert not td.keep_task_alive(1, 1000) and not td.keep_task_alive(14999, 1000)
    assert not td.keep_task_alive(10 ** 9, 1000)  # beyond every page
    # the short leases expire into zombies: not renewable, swept by their servants' next reports
    td.clock_advance_ms(1000)
    td.on_expiration_timer()
    assert not td.keep_task_alive(3500, 1000)
    zombies = {str(i): t for i, t in td.dump_internals()["tasks"].items() if t["zombie"]}
    assert len(zombies) == 999 and "3000" not in zombies  # (3000 was renewed above)
    by_loc = {}
    for i, t in zombies.items():
        by_loc.setdefault(t["servant_location"], []).append(int(i))
    for loc, mine in by_loc.items():
        still = mine[:2]  # the servant still lists two of them: those stay
        unknown = td.notify_servant_running_tasks(loc, still)
        assert unknown == still  # zombies are not "permitted": reported back (task_dispatcher.cc:256-262)
    left = td.dump_internals()
    assert sum(1 for t in left["tasks"].values() if t["zombie"]) == 2 * len(by_loc)
    # new grants after the churn: ids go on, freed pages are reused
    st, got, _ = td.wait_for_starting_new_tasks(["172.16.9.9"] * 5000, [digest] * 5000, [0] * 5000,
                                                expires_in_ms=10_000_000)
    assert (st == D.GRANTED).all() and [int(x) for x in got] == list(range(15000, 20000))
    td.free_tasks([int(x) for x in got])
    td.free_tasks(sorted(int(i) for i in left["tasks"]))
    assert td.dump_internals()["running_tasks"] == 0 and td.dump_internals()["tasks"] == {}
    # expiry of a servant frees its orphans
    st, got, locs = td.wait_for_starting_new_tasks(["172.16.9.9"] * 300, [digest] * 300, [0] * 300,
                                                   expires_in_ms=10_000_000)
    assert (st == D.GRANTED).all()
    td.clock_advance_ms(20_000_000)
    td.on_expiration_timer()
    d = td.dump_internals()
    assert d["servants_up"] == 0 and d["tasks"] == {}
    td.close()


def address_forms(make):
    """IsNetworkAddressEqual (task_dispatcher.cc:66-69): `location` starts with the requestor
    address followed by ':'. A location without ':' matches nobody — not even an empty
    requestor address — and ':port' matches the empty address."""
    td = make()
    ref = R.RefDispatcher() if R.available() else None
    for d in (td, ref):
        if d is None:
            continue
        d.keep_servant_alive("nocolon", ["d"], 1, 8, 0, memory_available=G50)
        d.keep_servant_alive(":8335", ["d"], 1, 8, 0, memory_available=G50)
    want = ["nocolon", ":8335"]  # "" avoids ":8335" (its own host), takes it last
    for k in range(2):
        st, _, loc = td.wait_for_starting_new_task("", "d")
        assert (st, loc) == (D.GRANTED, want[k])
        if ref is not None:
            rst, _, rloc = ref.wait_for_starting_new_task("", "d")
            assert (rst, rloc) == (R.OK, loc)
    td.close()
    if ref is not None:
        ref.close()


def address_prefix_forms(make, seed=5):
    """Locations with several ':' answer to EVERY prefix that ends right before one of them
    (task_dispatcher.cc:66-69): "[::1]:8335" — what scheduler_service_impl.cc:102-103 builds for
    an IPv6 peer — to "[::1]", but also to "[:" and "[". First the fixed cases, then a random
    stream of requests from such prefixes (and near misses) with heartbeats, frees and an expiry
    in between, answer for answer against the reference class."""
    td = make()
    ref = R.RefDispatcher()
    both = (ref, td)

    def ask(ip, digest="d"):
        rst, rid, rloc = ref.wait_for_starting_new_task(ip, digest)
        st, tid, loc = td.wait_for_starting_new_task(ip, digest)
        assert (int(st), loc or None) == (rst, rloc), (ip, st, loc, rst, rloc)
        return rst, rid, rloc

    for d in both:
        d.keep_servant_alive("[::1]:8335", ["d"], 2, 8, 0, memory_available=G50)
        d.keep_servant_alive("[::2]:8335", ["d"], 2, 8, 1, memory_available=G50)
    assert ask("[::1]")[2] == "[::2]:8335"   # its own servant is avoided although it is emptier
    assert ask("[::2]")[2] == "[::1]:8335"
    assert ask("::1")[2] == "[::1]:8335"     # what EndpointGetIp yields: matches nobody
    for d in both:
        d.keep_servant_alive("a:b:1", ["d"], 3, 8, 0, memory_available=G50)
        d.keep_servant_alive("a:c:2", ["d"], 3, 8, 0, memory_available=G50)
        d.keep_servant_alive("a:9", ["d"], 3, 8, 0, memory_available=G50)
    ids = []
    for ip in ["a:b", "a", "[", "a:c", "[:", "a:b:1", "a:", "[::1", "", "a"]:
        rst, rid, _ = ask(ip)
        if rst == R.OK:
            ids.append(rid)
    # random stream
    rng = np.random.default_rng(seed)
    locs = ["[::1]:8335", "[::2]:8335", "[::2]:9000", "a:b:1", "a:c:2", "a:9", "b:b:b:b", "b:b:7",
            "10.0.0.1:8335", "10.0.0.1:8336", ":5", "::6", "plain"]
    ips = sorted({loc[:i] for loc in locs for i in range(len(loc) + 1)})  # every prefix, matching or not
    live = list(ids)
    for step in range(400):
        ev = rng.random()
        if ev < 0.2:
            loc = locs[int(rng.integers(len(locs)))]
            kw = dict(max_tasks=int(rng.integers(0, 4)), num_processors=8,
                      current_load=int(rng.integers(0, 6)), memory_available=G50,
                      expires_in_ms=int(rng.integers(2000, 9000)))
            envs = ["d"] if rng.random() < 0.9 else ["e"]
            for d in both:
                d.keep_servant_alive(loc, envs, **kw)
        elif ev < 0.75:
            rst, rid, _ = ask(ips[int(rng.integers(len(ips)))], "d" if rng.random() < 0.9 else "e")
            if rst == R.OK:
                live.append(rid)
        elif ev < 0.93 and live:
            tid = live.pop(int(rng.integers(len(live))))
            for d in both:
                d.free_task(tid)
        else:
            ms = int(rng.integers(500, 3000))
            R.clock_advance_ms(ms)
            td.clock_advance_ms(ms)
            R.fire_timers()
            td.on_expiration_timer()
    assert_same_dump(td.dump_internals(), ref.dump_internals())
    ref.close()
    td.close()


def _random_personality(rng, i, digests):
    k = len(digests)
    if k <= 8:
        envs = [d for d in digests if rng.random() < 0.6] or [digests[int(rng.integers(k))]]
    else:  # many digests: every servant advertises a few of them
        envs = [digests[int(j)] for j in rng.choice(k, size=int(rng.integers(1, 6)), replace=False)]
    nproc = int(rng.choice([8, 16, 32, 64]))
    ded = rng.random() < 0.3
    return dict(
        location="10.1.%d.%d:%d" % (i // 200, i % 200, 8335 + (i % 3 == 0) * (i % 7)),
        envs=envs, max_tasks=0 if rng.random() < 0.05 else (nproc * (95 if ded else 40)) // 100,
        num_processors=nproc, current_load=int(rng.integers(0, int(nproc * 1.25))),
        priority=1 if ded else 2, version=int(rng.choice([19, 20, 20, 20])),
        total_memory=0 if rng.random() < 0.2 else 64 << 30,
        memory_available=(1 << 30) if rng.random() < 0.07 else (32 << 30),
        expires_in_ms=int(rng.integers(1500, 8000)))


def event_stream_matches_reference(make, seed, n_digests=3, n_pool=60, steps=400):
    digests = ["c0ffee%04d" % i + "0" * 54 for i in range(n_digests)]
    rng = np.random.default_rng(seed)
    ref = R.RefDispatcher()
    td = make()
    # some servants share a host (same ip, different port) to exercise the `self` rule
    hosts = ["10.1.%d.%d" % (i // 250, (i if i % 5 else max(i - 1, 0)) % 250) for i in range(n_pool)]
    live = {}          # task id -> location
    by_servant = {}    # location -> list of task ids ever granted there
    pick = lambda: digests[int(rng.integers(n_digests))]

    def heartbeat(i):
        p = _random_personality(rng, i, digests)
        p["location"] = "%s:%d" % (hosts[i], 9000 + i)
        for d in (ref, td):
            d.keep_servant_alive(**p)

    for i in range(0, n_pool, 2):
        heartbeat(i)
    for step in range(steps):
        ev = rng.random()
        if ev < 0.25:
            heartbeat(int(rng.integers(n_pool)))
        elif ev < 0.55:
            n = int(rng.integers(1, 40))
            ips = [hosts[int(rng.integers(n_pool))] if rng.random() < 0.3 else "172.16.0.%d" % rng.integers(250)
                   for _ in range(n)]
            dg = [("unknown" if rng.random() < 0.03 else pick()) for _ in range(n)]
            mv = [int(rng.choice([0, 20])) for _ in range(n)]
            lease = int(rng.integers(500, 6000))
            st, ids, locs = td.wait_for_starting_new_tasks(ips, dg, mv, expires_in_ms=lease)
            for k in range(n):
                rst, rid, rloc = ref.wait_for_starting_new_task(ips[k], dg[k], min_version=mv[k],
                                                                expires_in_ms=lease)
                assert (int(st[k]), locs[k] or None) == (rst, rloc), (seed, step, k)
                if rst == R.OK:
                    assert int(ids[k]) == rid
                    live[rid] = rloc
                    by_servant.setdefault(rloc, []).append(rid)
        elif ev < 0.70 and live:
            tid = int(rng.choice(list(live))) if rng.random() < 0.9 else 10 ** 9
            for d in (ref, td):
                d.free_task(tid)
            live.pop(tid, None)
        elif ev < 0.78 and live:
            tid = int(rng.choice(list(live))) if rng.random() < 0.9 else 10 ** 9
            ms = int(rng.integers(500, 6000))
            assert ref.keep_task_alive(tid, ms) == td.keep_task_alive(tid, ms)
        elif ev < 0.90 and by_servant:
            loc = str(rng.choice(list(by_servant)))
            ids = [t for t in by_servant[loc] if rng.random() < 0.7][-30:] + (
                [10 ** 9 + step] if rng.random() < 0.3 else [])
            a = ref.notify_servant_running_tasks(loc, ids)
            b = td.notify_servant_running_tasks(loc, ids)
            assert a == b, (seed, step)
            assert sorted(ref.get_running_tasks()) == sorted(td.get_running_tasks())
            # the copy-free view of the same snapshot (ydc_td_running_tasks_acquire): same entries,
            # every entry of a servant carrying that servant's location
            view = td.running_tasks_view()
            assert sorted((a_, b_) for a_, b_, _, _ in view) == sorted(td.get_running_tasks())
            assert sorted((a_, b_, l_) for a_, b_, l_, _ in view) == sorted(td.get_running_tasks(with_strings=True))
        else:
            ms = int(rng.integers(200, 2500))
            R.clock_advance_ms(ms)
            td.clock_advance_ms(ms)
            R.fire_timers()
            td.on_expiration_timer()
    # drain: everything that is still placeable must go to the same servants
    ips = ["172.16.9.9"] * 300
    dg = [digests[i % n_digests] for i in range(300)]
    st, ids, locs = td.wait_for_starting_new_tasks(ips, dg, [0] * 300)
    for k in range(300):
        rst, rid, rloc = ref.wait_for_starting_new_task(ips[k], dg[k], min_version=0)
        assert (int(st[k]), locs[k] or None) == (rst, rloc), (seed, "drain", k)
    dump = td.dump_internals()
    assert_same_dump(dump, ref.dump_internals())
    ref.close()
    td.close()
    return dump


TIME_KEYS = ("discovered_at", "expires_at", "started_at")


def assert_same_dump(ours, theirs):
    """DumpInternals (task_dispatcher.cc:538-614) key for key and value for value against the
    reference's own dump (written by the recording Json::Value stand-in of oracle/shims).
    Formatted wall-clock times are left out (the two clocks have different origins); ours has
    one extra key, "gpu"."""
    ours = dict(ours)
    ours.pop("gpu", None)

    def strip(d):
        return {k: v for k, v in d.items() if k not in TIME_KEYS}

    assert set(ours) == set(theirs) | ({"tasks"} if "tasks" not in theirs else set()) | (
        {"servants"} if "servants" not in theirs else set())
    for k in ("servants_up", "running_tasks", "capacity", "capacity_available",
              "capacity_unavailable"):
        assert ours[k] == theirs[k], k
    a, b = ours.get("servants", []), theirs.get("servants") or []
    assert len(a) == len(b)
    for x, y in zip(a, b):  # registry order
        assert set(x) == set(y), (sorted(x), sorted(y))
        assert strip(x) == strip(y)
    ta, tb = ours.get("tasks", {}), theirs.get("tasks") or {}
    assert set(ta) == set(tb)
    for k in ta:
        assert set(ta[k]) == set(tb[k])
        assert strip(ta[k]) == strip(tb[k]), k


# ---------------------------------------------------------------------------------------------
# Concurrent callers: linearizability against the reference
# ---------------------------------------------------------------------------------------------
def replay_oplog_through_reference(log, ref):
    """Replays the order in which calls took effect (ydc_td_oplog_take) through the reference
    class, one call at a time, each with the clock reading the concurrent run used; every answer
    must be the one the concurrent run gave (task_dispatcher.cc:93-140 WaitForStartingNewTask,
    :142-163 KeepTaskAlive, :165-188 FreeTask, :190-220 KeepServantAlive, :222-277
    NotifyServantRunningTasks, :498-536 OnExpirationTimer). Returns the number of records."""
    L = R.lib()
    base = None  # the reference's fake clock at the run's clock 0 (whole milliseconds on both sides)

    def set_clock(now_ns):
        nonlocal base
        assert now_ns % 1_000_000 == 0
        if base is None:
            base = L.ref_clock_now_ns() - now_ns
        delta = base + now_ns - L.ref_clock_now_ns()
        assert delta % 1_000_000 == 0
        R.clock_advance_ms(delta // 1_000_000)

    for k, e in enumerate(log):
        op = e["op"]
        if "now" in e:
            set_clock(e["now"])
        if op == "wait":
            st, tid, loc = ref.wait_for_starting_new_task(e["ip"], e["digest"], min_version=e["minv"],
                                                          expires_in_ms=e["lease"] // 1_000_000, timeout_in_ms=0)
            assert (st, tid, loc) == (e["st"], e.get("id"), e.get("loc")), (k, e, (st, tid, loc))
        elif op == "free":
            ref.free_task(e["id"])
        elif op == "renew":
            assert ref.keep_task_alive(e["id"], e["lease"] // 1_000_000) == bool(e["ok"]), (k, e)
        elif op == "servant":
            ref.keep_servant_alive(e["location"], e["envs"], e["max_tasks"], e["num_processors"], e["current_load"],
                                   priority=e["priority"], version=e["version"], total_memory=e["total_memory"],
                                   memory_available=e["memory_available"], expires_in_ms=e["lease"] // 1_000_000,
                                   reported=e["reported"], reason=e["reason"])
        elif op == "report":
            assert ref.notify_servant_running_tasks(e["loc"], e["ids"]) == e["unknown"], (k, e)
        elif op == "timer":
            R.fire_timers()
        else:
            raise AssertionError(e)
    return len(log)


def verify_linearizable(log, dump, threads, check_real_time=True):
    """log: the order in which calls took effect (ydc_td_oplog_take); dump: the final
    DumpInternals; threads: [{"ip": the requestor address the thread used (None: it only freed),
    "ops": [("wait", (status, id, location), t_invoke, t_return) | ("free", id, t_invoke,
    t_return), ...] in program order}]. Checks (1) program order, (2) real-time order, (3) the
    reference's answers and final state for the logged order. Returns the number of records."""
    waits_of, free_at = {}, {}
    for pos, e in enumerate(log):
        if e["op"] == "wait":
            waits_of.setdefault(e["ip"], []).append((pos, e))
        elif e["op"] == "free":
            free_at.setdefault(e["id"], []).append(pos)
    users = {}
    for t in threads:
        if t["ip"] is not None:
            users[t["ip"]] = users.get(t["ip"], 0) + 1
    intervals = []  # (log position, t_invoke, t_return)
    for t in threads:
        if t["ip"] is not None and users[t["ip"]] > 1:
            continue  # (two callers drew the same servant host: their entries cannot be told apart)
        last = -1
        calls_, cur = [], None  # attempts of one call: "try" restarts at 1
        for pos, e in waits_of.get(t["ip"], []):
            if e["try"] == 1:
                cur = []
                calls_.append(cur)
            cur.append((pos, e))
        it = iter(calls_)
        for kind, payload, t0, t1 in t["ops"]:
            if kind == "wait":
                attempts = next(it)
                pos, e = attempts[-1]
                assert (e["st"], e.get("id"), e.get("loc")) == tuple(payload), (t["ip"], payload, e)
                assert all(a["st"] == D.TIMEOUT for _, a in attempts[:-1])
                exact = len(attempts) == 1 or payload[0] != D.TIMEOUT
            else:
                # (a grant that did not fit the caller's buffer is freed by the library itself first)
                pos = free_at[payload][-1]
                exact = True
            assert pos > last, ("program order", t["ip"], kind, payload, pos, last)
            if exact:
                intervals.append((pos, t0, t1))
            last = pos
        assert next(it, None) is None, ("calls in the log that the thread never made", t["ip"])
    if check_real_time:
        latest_invoke = -1.0
        for pos, t0, t1 in sorted(intervals):
            assert t1 >= latest_invoke, ("took effect after a call that was made after it had returned", pos)
            latest_invoke = max(latest_invoke, t0)
    ref = R.RefDispatcher()
    try:
        n = replay_oplog_through_reference(log, ref)
        assert_same_dump(dump, ref.dump_internals())
    finally:
        ref.close()
    return n


def native_linearize(binary, out_path, n_servants=2000, n_threads=12, calls=2000, cap=0, seed=1, env=None):
    """Runs tests/native/td_linearize (threads in C++, through the C-ABI) and verifies what it wrote."""
    import json
    import subprocess
    r = subprocess.run([binary, out_path, str(n_servants), str(n_threads), str(calls), str(cap), str(seed)],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "TD-LINEARIZE-WRITTEN" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    j = json.load(open(out_path))
    threads = [{"ip": t["ip"], "ops": [(o[0], tuple(o[1:4]), o[4], o[5]) if o[0] == "wait" else (o[0], o[1], o[2], o[3])
                                       for o in t["ops"]]} for t in j["threads"]]
    n = verify_linearizable(j["log"], j["dump"], threads)
    kinds = {}
    for e in j["log"]:
        k = e["op"] + (":%d" % e["st"] if e["op"] == "wait" else "")
        kinds[k] = kinds.get(k, 0) + 1
    return {"records": n, "kinds": kinds, "requests_per_device_turn": j["requests_per_device_turn"],
            "retried_attempts": sum(1 for e in j["log"] if e["op"] == "wait" and e["try"] > 1)}


def concurrent_callers_linearize(make, seed=1, n_servants=2000, n_threads=12, calls=400, check_real_time=True,
                                 max_tasks_cap=None):
    """Many threads at once on one dispatcher — single requests (some parked with a deadline), small
    batches, FreeTask of own and of other threads' grants, lease renewals, heartbeats that change
    loads and capacities, servant reports, clock + timer ticks with leases and servants expiring —
    and then the proof that the combined result is what SOME sequential order of the same calls
    gives on the reference: the dispatcher logs the order in which the calls took effect; that
    order, replayed single-threaded through the reference's own translation units, must give
    identical (status, task id, location) for every call and an identical final DumpInternals.
    On top: every thread finds its own calls in the log in the order it made them (a FreeTask
    before the thread's next request included), and — check_real_time — a call that had returned
    before another was made precedes it in the log."""
    import threading
    import time
    rng0 = np.random.default_rng(seed)
    digests = ["c0ffee%04d" % i + "0" * 54 for i in range(4)]
    td = make(fake_clock=True)
    pers = []
    for i in range(n_servants):
        p = _random_personality(rng0, i, digests)
        p["location"] = "10.%d.%d.%d:%d" % (2 + i // 60000, (i // 250) % 240, i % 250, 8335)
        p["expires_in_ms"] = int(rng0.integers(3000, 9000)) if i % 7 == 0 else 3_600_000
        if max_tasks_cap is not None:  # a pool that saturates: requests park, FreeTask wakes them
            p["max_tasks"] = min(p["max_tasks"], max_tasks_cap)
        pers.append(p)
    td.oplog_enable(True)
    for p in pers:
        td.keep_servant_alive(**p)
    stop = threading.Event()
    shared_lock = threading.Lock()
    handed_over = []       # grants left for the freer thread
    errors = []
    per_thread = {}        # thread key -> [(kind, payload, t_invoke, t_return), ...] in program order

    def guard(fn):
        def run(*a):
            try:
                fn(*a)
            except BaseException as e:  # noqa: BLE001
                errors.append(repr(e))
                stop.set()
        return run

    @guard
    def caller(k):
        rng = np.random.default_rng(1000 * seed + k)
        ip = "172.20.%d.7" % k if k % 3 else pers[int(rng.integers(n_servants))]["location"].split(":")[0]
        mine, hist = [], per_thread.setdefault(("caller", k), [])
        hist.append(("ip", ip, 0, 0))
        for _ in range(calls):
            if stop.is_set():
                return
            ev = rng.random()
            if ev < 0.62:
                dg = "unknown" if rng.random() < 0.02 else digests[int(rng.integers(4))]
                mv = int(rng.choice([0, 20]))
                lease = int(rng.choice([40, 200, 60000]))
                timeout = int(rng.choice([0, 0, 0, 3]))
                t0 = time.perf_counter()
                st, tid, loc = td.wait_for_starting_new_task(ip, dg, min_version=mv, expires_in_ms=lease,
                                                             timeout_in_ms=timeout)
                hist.append(("wait", (st, tid, loc), t0, time.perf_counter()))
                if st == D.GRANTED:
                    mine.append(tid)
            elif ev < 0.70:
                n = int(rng.integers(2, 9))
                dg = digests[int(rng.integers(4))]
                t0 = time.perf_counter()
                st, ids, locs = td.wait_for_starting_new_tasks([ip] * n, [dg] * n, [0] * n, expires_in_ms=60000)
                t1 = time.perf_counter()
                for j in range(n):
                    ok = int(st[j]) == D.GRANTED
                    hist.append(("wait", (int(st[j]), int(ids[j]) if ok else None, locs[j] if ok else None), t0, t1))
                    if ok:
                        mine.append(int(ids[j]))
            elif ev < 0.90 and mine:
                tid = mine.pop(int(rng.integers(len(mine))))
                if rng.random() < 0.25:
                    with shared_lock:
                        handed_over.append(tid)
                else:
                    t0 = time.perf_counter()
                    td.free_task(tid)
                    hist.append(("free", tid, t0, time.perf_counter()))
            elif mine:
                tid = mine[int(rng.integers(len(mine)))]
                td.keep_task_alive(tid, int(rng.choice([50, 5000])))
        with shared_lock:
            handed_over.extend(mine)

    @guard
    def freer():
        hist = per_thread.setdefault(("freer", 0), [])
        while not stop.is_set() or handed_over:
            with shared_lock:
                tid = handed_over.pop() if handed_over else None
            if tid is None:
                time.sleep(0.0005)
                continue
            t0 = time.perf_counter()
            td.free_task(tid)
            hist.append(("free", tid, t0, time.perf_counter()))

    @guard
    def registry():
        rng = np.random.default_rng(77 + seed)
        while not stop.is_set():
            i = int(rng.integers(n_servants))
            p = dict(pers[i])
            if rng.random() < 0.8:  # the daemon's next report: a new load figure
                p["current_load"] = int(rng.integers(0, int(p["num_processors"] * 1.25)))
            else:                   # ... or a changed machine (capacity, environments)
                q = _random_personality(rng, i, digests)
                p.update(envs=q["envs"], max_tasks=q["max_tasks"], num_processors=q["num_processors"],
                         version=q["version"], memory_available=q["memory_available"])
                if max_tasks_cap is not None:
                    p["max_tasks"] = min(p["max_tasks"], max_tasks_cap)
            pers[i] = p
            td.keep_servant_alive(**p)
            if rng.random() < 0.3:
                td.notify_servant_running_tasks(p["location"], [int(x) for x in rng.integers(0, 4000, size=5)])
            time.sleep(0.0002)

    @guard
    def clock():
        tick = 0
        while not stop.is_set():
            td.clock_advance_ms(1)
            tick += 1
            if tick % 25 == 0:
                td.on_expiration_timer()
            time.sleep(0.0005)

    callers = [threading.Thread(target=caller, args=(k,)) for k in range(n_threads)]
    others = [threading.Thread(target=f) for f in (freer, registry, clock)]
    [t.start() for t in others + callers]
    [t.join() for t in callers]
    stop.set()
    [t.join() for t in others]
    assert not errors, errors[:3]
    log = td.oplog_take()
    td.oplog_enable(False)
    dump = td.dump_internals()
    hs = td.host_stats()

    threads = []
    for key, hist in per_thread.items():
        if key[0] == "caller":
            threads.append({"ip": hist[0][1], "ops": [(k, pl, t0, t1) for k, pl, t0, t1 in hist[1:]]})
        else:
            threads.append({"ip": None, "ops": list(hist)})
    n = verify_linearizable(log, dump, threads, check_real_time)
    td.close()
    kinds = {}
    for e in log:
        k = e["op"] + (":%d" % e["st"] if e["op"] == "wait" else "")
        kinds[k] = kinds.get(k, 0) + 1
    parked = sum(1 for e in log if e["op"] == "wait" and e["try"] > 1)
    return {"records": n, "kinds": kinds, "retried_attempts": parked, "threads": n_threads,
            "requests_per_device_turn": hs["requests"] / max(hs["batches"], 1)}


def event_stream_at_scale(make, seed, n_pool=2000, prefill=50_000, steps=20_000, n_digests=4, batch_sizes=(
        1, 1, 2, 3, 4, 8, 16, 32, 64, 128, 256)):
    """The differential of event_stream_matches_reference at the scale the scheduler runs at:
    2000 servants, >= 5*10^4 live leases throughout, >= 2*10^4 events whose grant batches have
    1 .. 256 requests — so that within ONE stream the resident tick kernel, the launched tick and
    the batch pipeline all serve calls, between structural heartbeats (new / changed / expired
    servants: class changes, ydc_remove_servants), bulk frees (COMMIT / release) and timer ticks
    that turn leases into zombies — every answer compared with the reference class's."""
    digests = ["c0ffee%04d" % i + "0" * 54 for i in range(n_digests)]
    rng = np.random.default_rng(seed)
    ref = R.RefDispatcher()
    td = make()
    hosts = ["10.%d.%d.%d" % (1 + i // 62500, (i // 250) % 250, (i if i % 5 else max(i - 1, 0)) % 250) for i in range(n_pool)]
    live_ids, live_pos = [], {}
    by_servant = {}
    pick = lambda: digests[int(rng.integers(n_digests))]

    def add_live(tid, loc):
        live_pos[tid] = len(live_ids)
        live_ids.append(tid)
        by_servant.setdefault(loc, []).append(tid)

    def drop_live(tid):
        at = live_pos.pop(tid, None)
        if at is None:
            return
        last = live_ids.pop()
        if last != tid:
            live_ids[at] = last
            live_pos[last] = at

    def heartbeat(i, long_lived=False):
        p = _random_personality(rng, i, digests)
        p["location"] = "%s:%d" % (hosts[i], 9000 + i % 50)
        p["num_processors"] *= 4
        p["max_tasks"] *= 4
        p["current_load"] *= 4
        if long_lived or rng.random() < 0.8:
            p["expires_in_ms"] = 3_600_000
        for d in (ref, td):
            d.keep_servant_alive(**p)

    def grant_batch(n, lease):
        ips = [hosts[int(rng.integers(n_pool))] if rng.random() < 0.2 else "172.16.%d.%d" % (rng.integers(250), rng.integers(250))
               for _ in range(n)]
        dg = [("unknown" if rng.random() < 0.01 else pick()) for _ in range(n)]
        mv = [int(rng.choice([0, 20])) for _ in range(n)]
        st, ids, locs = td.wait_for_starting_new_tasks(ips, dg, mv, expires_in_ms=lease)
        for k in range(n):
            rst, rid, rloc = ref.wait_for_starting_new_task(ips[k], dg[k], min_version=mv[k], expires_in_ms=lease)
            assert (int(st[k]), locs[k] or None) == (rst, rloc), (seed, "batch of", n, k)
            if rst == R.OK:
                assert int(ids[k]) == rid
                add_live(rid, rloc)

    for i in range(n_pool):
        if i % 10:
            heartbeat(i, long_lived=True)
    while len(live_ids) < prefill:
        before = len(live_ids)
        grant_batch(5000, 3_600_000)
        assert len(live_ids) > before, "the pool cannot hold the prefill"
    paths = td.dump_internals().get("gpu", {})
    low_water = len(live_ids)
    for step in range(steps):
        ev = rng.random()
        if ev < 0.22:
            heartbeat(int(rng.integers(n_pool)))
        elif ev < 0.52:
            grant_batch(int(rng.choice(batch_sizes)), 3_600_000 if rng.random() < 0.8 else int(rng.integers(500, 6000)))
        elif ev < 0.72 and live_ids:
            k = int(rng.integers(1, 200))
            ids = [live_ids[int(j)] for j in rng.integers(0, len(live_ids), size=k)]
            ids = list(dict.fromkeys(ids)) + ([10 ** 9] if rng.random() < 0.05 else [])
            if rng.random() < 0.5:
                for d in (ref, td):
                    d.free_tasks(ids)
            else:
                for t in ids[:8]:
                    for d in (ref, td):
                        d.free_task(t)
                ids = ids[:8]
            # (free_tasks = one FreeTask per id on both sides: an unknown id ends only its own call)
            for t in ids:
                drop_live(t)
        elif ev < 0.78 and live_ids:
            tid = live_ids[int(rng.integers(len(live_ids)))] if rng.random() < 0.9 else 10 ** 9
            ms = int(rng.integers(500, 6000))
            assert ref.keep_task_alive(tid, ms) == td.keep_task_alive(tid, ms)
        elif ev < 0.90 and by_servant:
            locs = list(by_servant)
            loc = locs[int(rng.integers(len(locs)))]
            ids = [t for t in by_servant[loc][-40:] if rng.random() < 0.7] + ([10 ** 9 + step] if rng.random() < 0.3 else [])
            a = ref.notify_servant_running_tasks(loc, ids)
            b = td.notify_servant_running_tasks(loc, ids)
            assert a == b, (seed, step)
            for t in by_servant[loc][-40:]:
                if t not in ids:
                    pass  # (swept only if it was a zombie: the final dump comparison sees it)
        else:
            ms = int(rng.integers(200, 2500))
            R.clock_advance_ms(ms)
            td.clock_advance_ms(ms)
            R.fire_timers()
            td.on_expiration_timer()
        if step % 500 == 0:
            # leases that both sides dropped (zombies swept, orphans): resynchronise the model from the dump
            known = set(td.dump_internals()["tasks"])
            for t in [t for t in live_ids if str(t) not in known]:
                drop_live(t)
            if len(live_ids) < prefill:
                grant_batch(min(5000, prefill - len(live_ids) + 500), 3_600_000)
            low_water = min(low_water, len(live_ids))
    dump = td.dump_internals()
    assert_same_dump(dump, ref.dump_internals())
    stats = td.host_stats()
    ref.close()
    td.close()
    return {"live_low_water": low_water, "live_at_end": len(dump["tasks"]), "servants_at_end": dump["servants_up"],
            "requests": stats["requests"], "batches": stats["batches"], "gpu": dump.get("gpu", paths)}
