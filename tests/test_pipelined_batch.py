"""PipelinedBatchDemodulator (tetraear_amd/batch.py): plans of the same batch geometry taking the steps in turn -- each plan's
outputs are a lone plan's bit for bit, chunk k can be fed to plan k % depth, per-stage timing orders the steps on the device
(tdm_plan_wait_for)."""
import numpy as np
import pytest


def _valid_equal(a, b):
    """outputs (hard, soft, n_soft, best_phase, min_margin) equal over their valid parts"""
    np.testing.assert_array_equal(a[2], b[2])
    np.testing.assert_array_equal(a[3], b[3])
    np.testing.assert_array_equal(a[4], b[4])
    for r in range(len(a[2])):
        k = int(a[2][r])
        np.testing.assert_array_equal(a[0][r, :max(k - 1, 0)], b[0][r, :max(k - 1, 0)])
        np.testing.assert_array_equal(a[1][r, :k], b[1][r, :k])


@pytest.mark.gpu
def test_pipelined_plans_equal_a_lone_plan_and_take_chunks_in_turn():
    from tetraear_amd import synth
    from tetraear_amd.batch import BatchDemodulator, PipelinedBatchDemodulator, batch_demodulator
    n, rows = 65536, 9
    chunks = [np.concatenate([synth.noise_cu8(n, 8100 + 50 * c + r) for r in range(rows)]) for c in range(5)]
    foffs = np.linspace(-3000.0, 3000.0, rows)
    one = BatchDemodulator(2.4e6, n, rows, "cu8")
    one.alloc_device_io()
    refs = []
    for u8 in chunks:
        one.upload(u8, freq_offsets=foffs)
        one.enqueue()
        refs.append(one.download())
    one.close()
    for depth in (2, 3):
        pl = PipelinedBatchDemodulator(2.4e6, n, rows, "cu8", depth=depth)
        assert pl.depth == depth
        pl.alloc_device_io()
        # one resident batch, many steps: every plan ends with the lone plan's outputs
        pl.upload(chunks[0], freq_offsets=foffs)
        for _ in range(2 * depth + 1):
            pl.enqueue()
        for o in pl.download_all():
            _valid_equal(refs[0], o)
        _valid_equal(refs[0], pl.download())
        # a capture loop: chunk k to plan k % depth, `depth` chunks in flight, outputs collected a round later
        pl._turn = 0
        got = {}
        for k, u8 in enumerate(chunks):
            if k >= depth:
                got[k - depth] = pl.plans[k % depth].download()     # (the plan about to be reused: its previous chunk's outputs)
            pl.upload(u8, freq_offsets=foffs, slot=k)
            pl.enqueue()
        for k in range(len(chunks) - depth, len(chunks)):
            got[k] = pl.plans[k % depth].download()
        for k in range(len(chunks)):
            _valid_equal(refs[k], got[k])
        # the per-stage pass: steps one after the other on the device, same outputs
        pl.upload(chunks[1], freq_offsets=foffs)
        pl.time_begin(per_stage=True)
        for _ in range(depth):
            pl.enqueue()
        ms = pl.time_end()
        st = pl.stage_times()
        assert ms > 0 and st["dec_block"] > 0
        for o in pl.download_all():
            _valid_equal(refs[1], o)
        pl.close()
    assert isinstance(batch_demodulator(2.4e6, n, 4, "cu8", depth=1), BatchDemodulator)
    auto = batch_demodulator(2.4e6, n, 4, "cu8")
    assert isinstance(auto, PipelinedBatchDemodulator) and auto.depth == 3
    auto.close()


@pytest.mark.gpu
def test_plan_wait_for_orders_two_plans_on_the_device():
    """tdm_plan_wait_for: plan B's pass enqueued behind plan A's with the wait runs alone (its own event span is a lone pass's),
    without it the two share the device and B's span grows."""
    from tetraear_amd import _lib, synth
    from tetraear_amd.batch import BatchDemodulator
    n, rows = 262144, 64
    u8 = synth.noise_cu8(n * rows, 8200)
    a = BatchDemodulator(2.4e6, n, rows, "cu8")
    b = BatchDemodulator(2.4e6, n, rows, "cu8")
    for p in (a, b):
        p.alloc_device_io()
        p.upload(u8)
        for _ in range(20):
            p.enqueue()
        p.sync()
    a.time_begin(per_stage=False); a.enqueue(); t_alone = a.time_end()
    a.enqueue()
    b.wait_for(a)
    b.time_begin(per_stage=False); b.enqueue(); t_ordered = b.time_end()
    a.sync()
    a.enqueue()
    b.time_begin(per_stage=False); b.enqueue(); t_beside = b.time_end()
    a.sync()
    print(f"one pass alone {t_alone:.4f} ms, behind the other plan's pass (tdm_plan_wait_for) {t_ordered:.4f} ms, beside it {t_beside:.4f} ms")
    assert t_ordered < 1.25 * t_alone
    with pytest.raises(_lib.TetraHipError):
        _lib.check(_lib.load().tdm_plan_wait_for(None, a.handle))
    a.close(); b.close()


def test_pipelined_host_logic_with_stand_in_plans(monkeypatch):
    """CPU tier: the turn-taking logic of PipelinedBatchDemodulator on stand-in plans (no device): step k goes to plan k % depth,
    `upload(slot=k)` feeds that plan only, `download` returns the last step's plan, per-stage timing orders every step behind the
    previous one (wait_for) and plain timing does not."""
    import tetraear_amd.batch as batch
    log = []

    class Plan:
        count = 0

        def __init__(self, *a, **k):
            self.idx = Plan.count
            Plan.count += 1
            self.info, self.soft_dtype, self.data = "info", np.complex128, None

        def alloc_device_io(self, shared_input=False): log.append(("alloc", self.idx, shared_input))
        def upload(self, iq, fo=None, ps=None): self.data = iq
        def enqueue(self): log.append(("enqueue", self.idx))
        def wait_for(self, other): log.append(("wait", self.idx, other.idx))
        def sync(self): pass
        def download(self): return ("out", self.idx, self.data)
        def time_begin(self, per_stage=True): log.append(("begin", self.idx, per_stage))
        def time_end(self): return 1.0 + self.idx
        def stage_times(self): return {"dec_block": 0.1 * (self.idx + 1)}
        def close(self): log.append(("close", self.idx))

    monkeypatch.setattr(batch, "BatchDemodulator", Plan)
    pl = batch.PipelinedBatchDemodulator(2.4e6, 4096, 4, "cu8", depth=3)
    assert pl.depth == 3
    pl.alloc_device_io()
    pl.upload("all")
    assert [p.data for p in pl.plans] == ["all"] * 3
    for k in range(5):
        pl.upload(f"chunk{k}", slot=k)
        pl.enqueue()
    assert [e[1] for e in log if e[0] == "enqueue"] == [0, 1, 2, 0, 1]
    assert not [e for e in log if e[0] == "wait"]                  # steps overlap unless a per-stage pass is on
    assert pl.download() == ("out", 1, "chunk4") and [o[2] for o in pl.download_all()] == ["chunk3", "chunk4", "chunk2"]
    pl.time_begin(per_stage=True)
    pl.enqueue(); pl.enqueue()
    assert [e for e in log if e[0] == "wait"] == [("wait", 2, 1), ("wait", 0, 2)]
    assert pl.time_end() == 3.0 and abs(pl.stage_times()["dec_block"] - 0.2) < 1e-12
    keep = Plan.stage_times
    Plan.stage_times = lambda self: {} if self.idx == 1 else keep(self)      # (a plan that took no step of a short per-stage pass)
    assert abs(pl.stage_times()["dec_block"] - 0.2) < 1e-12
    Plan.stage_times = keep
    pl.enqueue()
    assert len([e for e in log if e[0] == "wait"]) == 2             # (back to overlapping steps after time_end)
    pl.close()
    assert [e[1] for e in log if e[0] == "close"] == [0, 1, 2]
    assert isinstance(batch.batch_demodulator(2.4e6, 4096, 4, "cu8", depth=1), Plan)
