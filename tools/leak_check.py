"""Device-memory leak check (GPU box; uses torch only to read the device's free memory): 60 create / use / destroy cycles of every
kind of plan (pipelined reference-mode batch, host-pointer batch, time-batched plan option, TETRA / Gardner plans on cf32 and cu8,
SignalProcessor incl. the FFT resample) must leave the free memory where it was.  Round 6: delta 0.0 MB."""
import sys; sys.path.insert(0, ".")
import numpy as np, torch
from tetraear_amd.batch import BatchDemodulator, batch_demodulator
from tetraear_amd._lib import MODE_TETRA, MODE_TETRA_GARDNER
from tetraear_amd import synth
from tetraear_amd.signal import SignalProcessor
def free(): torch.cuda.synchronize(); return torch.cuda.mem_get_info()[0]
u8 = synth.noise_cu8(65536 * 8, 1)
x32 = (np.random.default_rng(0).standard_normal(8 * 8192) + 0j).astype(np.complex64)
def cycle():
    bd = batch_demodulator(2.4e6, 65536, 8, "cu8"); bd.alloc_device_io(); bd.upload(u8); bd.enqueue(); bd.download(); bd.close()
    b = BatchDemodulator(2.4e6, 65536, 8, "cu8"); b.process(u8, freq_offsets=[100.0] * 8); b.set_rows_per_chunk(4); b.close()
    for mode in (MODE_TETRA, MODE_TETRA_GARDNER):
        t = BatchDemodulator(72000.0, 8192, 8, "cf32", mode=mode); t.process(x32); t.close()
    t = BatchDemodulator(72000.0, 8192, 8, "cu8", mode=MODE_TETRA); t.process(u8[:8 * 8192 * 2]); t.close()
    p = SignalProcessor(2.4e6); p.process_cu8(u8[:2 * 65536], 50.0); p.resample(synth.cu8_to_c128(u8[:2 * 20000]), 1.2e6)
cycle(); cycle()
f0 = free()
for i in range(60): cycle()
f1 = free()
print("free before", f0, "after 60 cycles", f1, "delta MB", (f0 - f1) / 1e6)
