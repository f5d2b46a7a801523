"""Choosing a torch side stream that really runs beside the current stream (MI355X-native helper; no reference counterpart).

The two-stream schemes of this repo -- the occupancy march and the counting half of the hash-grid backward of step k+2 beside the
shading / backward / Adam of step k (`NeRFRenderer.march_train(plan_backward=True)`, `LAENeRF.plan_backward`, bench.py's
`grouped_pipeline`) -- need a side stream that (a) executes concurrently with the main stream and (b) hands work over to it quickly.
Neither is a given: torch hands its pooled streams out round-robin and HIP maps streams onto a few hardware queues.  Measured on
MI355X / ROCm 7.2 (round 5):
- a pool stream that lands on the main stream's hardware queue executes in line with it: the scheme silently loses its overlap
  (`style_step` 0.3205 / 0.3052 / 0.3204 ms in three runs of one process, depending on which pool stream the call got);
- once a HIGH-priority HIP stream exists in the process (this library's frame loop creates one for its lookahead), some pool streams
  pay ~60-130 us extra per cross-stream dependency (event recorded on one stream, waited for on the other: 86-157 against 30 us
  per round trip in the probe below); a grouped pipeline on such a stream ran at 0.86 instead of 0.64 ms per step (flower) and at
  0.78 instead of 0.31 (LAENeRF step), with unchanged kernel times;
- a high-priority torch side stream avoids the first problem but ran whole runs 3x slower when it was the first such stream of a
  process that had replayed other graphs -- so the priority stays the default.
- the more streams a process has put to use, the likelier the NEXT stream of another priority class is such a slow queue (the frame
  loop's own lookahead stream: 23 instead of 9.8 ms per frame once three other streams were in use; it now times its candidates
  too) -- so the probe stops at the first candidate that is good enough instead of trying them all.
`concurrent_side_stream()` probes pool streams until one runs beside the main stream with a prompt round trip (at most six) and
returns the best one seen."""
import os
import time

import torch

PROBES = []          # one record per call, newest last (bench.py puts the headline's into its line)
PROMPT_ROUND_TRIP_S = 45e-6      # event round trip main -> side -> main: 27-34 us on a prompt queue, 64-157 us on a slow one


def _spin(streams, cycles):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for st in streams:
        with torch.cuda.stream(st):
            torch.cuda._sleep(cycles)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def _round_trip(main, side, n=40):
    """seconds per main -> side -> main hand-over (event record + cross-stream wait each way, a ~20 us spin on either side)"""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        e1 = torch.cuda.Event()
        e1.record(main)
        side.wait_event(e1)
        with torch.cuda.stream(side):
            torch.cuda._sleep(2000)
            e2 = torch.cuda.Event()
            e2.record(side)
        main.wait_event(e2)
        torch.cuda._sleep(2000)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def concurrent_side_stream(candidates=6):
    """-> (stream, record): the first candidate that runs BESIDE the current stream (two ~0.4 ms single-thread spin kernels, one per
    stream, take one spin's time and not two) with a prompt cross-stream round trip (< 45 us), else the concurrent one with the
    shortest round trip; the last candidate if none runs beside it (e.g. under a serialising profiler) -- the record says so
    (`concurrent`)."""
    candidates = int(os.environ.get("LAE_STREAM_CANDIDATES", candidates))      # A/B switch
    if candidates <= 0:                                   # no probe: whatever torch hands out
        return torch.cuda.Stream(), {"candidates": 0, "concurrent": None}
    main = torch.cuda.current_stream()
    cycles = 100000
    _spin([main], cycles)
    solo = min(_spin([main], cycles) for _ in range(3))
    cycles = int(cycles * min(max(4e-4 / max(solo, 1e-6), 1.0), 1000.0))
    solo = min(_spin([main], cycles) for _ in range(3))
    seen = []
    for _ in range(candidates):
        side = torch.cuda.Stream()
        pair = min(_spin([main, side], cycles) for _ in range(3))
        hop = min(_round_trip(main, side) for _ in range(2))
        seen.append((pair < 1.5 * solo, hop, pair, side))
        if seen[-1][0] and hop < PROMPT_ROUND_TRIP_S:     # good enough: every further candidate put to use is one more hardware queue
            break                                         # in play for everybody else (the frame loop's side stream, see above)
    good = [c for c in seen if c[0]]
    beside, hop, pair, side = min(good, key=lambda c: c[1]) if good else seen[-1]
    rec = {"candidates": len(seen), "concurrent": bool(beside), "solo_ms": round(solo * 1e3, 3), "pair_ms": round(pair * 1e3, 3),
           "round_trip_us": round(hop * 1e6, 1), "round_trips_seen_us": [round(c[1] * 1e6, 1) for c in seen],
           "in_line_candidates": sum(1 for c in seen if not c[0])}
    PROBES.append(rec)
    return side, rec
