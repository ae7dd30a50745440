"""Timing discipline shared by bench.py and every leg of bench_legs.py.

A leg used to run a fixed handful of warm-up steps and ONE timed block; after a phase in which the GPU idled (the previous leg's CPU-side
oracle check) the first replays of a freshly instantiated hipGraph ran in a slow mode and a block of 8 steps caught it whole -- one driver
run reported half the rate of every other run.  Now every timed number is taken the same way:

  1. settle by TIME: replay the step (each step ends in the token read-back, i.e. a stream synchronisation) until at least `settle_ms`
     of GPU work have run AND three consecutive step times agree within `agree` (2 %); capped, and the line says whether it settled;
  2. `blocks` (>= 3) timed blocks of EXACTLY `steps` steps each, every block from the same start state (the caller's `reset`: the greedy
     loop is re-begun at the same context, then `warmup` untimed steps), bracketed by a device synchronisation on both sides;
  3. the reported value is the MEDIAN block; min, max and every block ride along, together with the library's capture / eager-step
     counters before and after the timed region (mi355_llama_graph_captures: a re-capture inside a timed block would be visible).
"""
import time


def settle(step, settle_ms=300.0, agree=0.02, max_steps=4000, max_ms=6000.0, reset=None, reset_every=0, fixed_steps=0):
    """run `step()` (one synchronising step) until >= settle_ms have elapsed and the last three step times agree within `agree`;
    with `reset_every` the loop is re-begun every that many steps, so that the settle phase never leaves the benchmark's context range"""
    ts, total = [], 0.0
    settled = False
    if fixed_steps:                       # several ranks must run the SAME number of steps (every step holds collectives): no local criterion
        max_steps, max_ms, settle_ms = fixed_steps, float("inf"), float("inf")
    while len(ts) < max_steps and total < max_ms:
        if reset and reset_every and ts and len(ts) % reset_every == 0:
            reset()
        t0 = time.perf_counter()
        step()
        dt = (time.perf_counter() - t0) * 1e3
        ts.append(dt)
        total += dt
        if total >= settle_ms and len(ts) >= 3:
            last = ts[-3:]
            if max(last) <= (1.0 + agree) * min(last):
                settled = True
                break
    return {"steps": len(ts), "ms": round(total, 1), "settled": settled,
            "first_steps_ms": [round(x, 3) for x in ts[:4]], "last_steps_ms": [round(x, 3) for x in ts[-3:]]}


def timed_blocks(step, sync, steps, warmup=0, blocks=3, reset=None, stats=None, settle_ms=300.0, fixed_settle_steps=0, reduce_max=None):
    """-> dict(median_s, min_s, max_s, blocks_ms_per_step, settle, counters).  `step()` = one step of the hot path incl. its
    read-back; `sync()` = device synchronisation; `reset()` puts the loop back to the benchmark's start state (untimed); `stats()` ->
    tuple of library counters (graph captures, eager steps); `fixed_settle_steps` / `reduce_max(seconds) -> seconds`: multi-rank runs
    settle by a fixed step count and take every block's maximum over the ranks."""
    if reset:
        reset()
    info = settle(step, settle_ms=settle_ms, reset=reset, reset_every=(steps + warmup) if reset else 0,
                  fixed_steps=fixed_settle_steps)
    c0 = stats() if stats else None
    dts = []
    for _ in range(max(1, blocks)):
        if reset:
            reset()
        for _ in range(warmup):
            step()
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        sync()
        dt = time.perf_counter() - t0
        dts.append(reduce_max(dt) if reduce_max else dt)
    c1 = stats() if stats else None
    srt = sorted(dts)
    out = {"median_s": srt[len(srt) // 2], "min_s": srt[0], "max_s": srt[-1],
           "blocks_ms_per_step": [round(1e3 * d / steps, 4) for d in dts], "settle": info}
    if c0 is not None:
        # reset() re-begins the loop at the same (batch, table width, ctx cap): no new shape, so both counters must stand still
        out["graph_captures_in_timed_region"] = c1[0] - c0[0]
        out["eager_steps_in_timed_region"] = c1[1] - c0[1]
        out["graph_captures_total"] = c1[0]
    return out


def timed_calls(call, sync, min_calls=3, settle_ms=300.0, max_settle_calls=50):
    """for steps that are long by themselves (prompt steps): settle by time, then >= 3 single timed calls; median / min / max seconds"""
    total, n = 0.0, 0
    while (total < settle_ms or n < 1) and n < max_settle_calls:
        t0 = time.perf_counter()
        call()
        sync()
        total += (time.perf_counter() - t0) * 1e3
        n += 1
    dts = []
    for _ in range(max(3, min_calls)):
        sync()
        t0 = time.perf_counter()
        call()
        sync()
        dts.append(time.perf_counter() - t0)
    srt = sorted(dts)
    return {"median_s": srt[len(srt) // 2], "min_s": srt[0], "max_s": srt[-1], "calls_ms": [round(1e3 * d, 3) for d in dts],
            "settle": {"calls": n, "ms": round(total, 1)}}


def spread_fields(tb, per, scale):
    """the (median, min, max) of a timed_blocks / timed_calls result as rates: per = units per block, scale as needed"""
    return {"value_median": round(per / tb["median_s"] * scale, 1), "value_min": round(per / tb["max_s"] * scale, 1),
            "value_max": round(per / tb["min_s"] * scale, 1)}
