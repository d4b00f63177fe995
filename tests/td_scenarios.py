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
