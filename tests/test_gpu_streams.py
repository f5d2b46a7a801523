"""laenerf_amd.streams.concurrent_side_stream: the side stream the two-stream schemes run on."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_concurrent_side_stream_runs_beside_the_current_stream():
    """the chosen stream is not the current one, runs beside it (two spin kernels take one spin's time) and its cross-stream
    round trip is the shortest of the candidates; work ordered through events on it arrives"""
    from laenerf_amd.streams import PROBES, concurrent_side_stream
    n0 = len(PROBES)
    side, rec = concurrent_side_stream(candidates=4)
    assert len(PROBES) == n0 + 1 and PROBES[-1] is rec
    main = torch.cuda.current_stream()
    assert side != main and 1 <= rec["candidates"] <= 4 and len(rec["round_trips_seen_us"]) == rec["candidates"]
    assert rec["concurrent"], rec                                   # an MI355X box runs two streams side by side
    assert rec["pair_ms"] < 1.5 * rec["solo_ms"]
    assert rec["round_trip_us"] <= min(rec["round_trips_seen_us"]) * 1.5 + 1e-6 or rec["in_line_candidates"] > 0
    x = torch.zeros(1024, device="cuda")
    side.wait_stream(main)
    with torch.cuda.stream(side):
        x += 1
    main.wait_stream(side)
    x += 1
    torch.cuda.synchronize()
    assert float(x.sum()) == 2048.0
