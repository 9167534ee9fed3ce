"""Batched cf32 file ingest: hd_host_iqfiles_* against a restatement of the reference's
IQSource_File<float>::get (code/IQSource/IQSource_File.h:124-172) -- std::ifstream end-of-file semantics included --
and, on the GPU, hd_ingest_run against feeding the same samples directly."""
import numpy as np
import pytest


class RefFile:
    """IQSource_File::get restated: the eof flag is set by the read that runs into the end; the NEXT call rewinds."""

    def __init__(self, data: np.ndarray, loop: bool):
        self.d, self.loop, self.pos, self.eof = data, loop, 0, False

    def get(self, want: int) -> np.ndarray:
        if self.eof:
            if not self.loop:
                return self.d[:0]
            self.eof, self.pos = False, 0
        n = min(want, len(self.d))
        out = self.d[self.pos:self.pos + n]
        if len(out) < n:
            self.eof = True                    # ifstream::read came up short: eofbit (and failbit)
        self.pos += len(out)
        return out


def ref_rounds(datas, loop, chunk, granule, rounds):
    files = [RefFile(d, loop) for d in datas]
    carry = [d[:0] for d in datas]
    out = []
    for _ in range(rounds):
        row, alive = [], 0
        for s, f in enumerate(files):
            got = f.get(chunk - len(carry[s]))
            alive += len(got) > 0
            tot = np.concatenate([carry[s], got])
            use = len(tot) - len(tot) % granule
            row.append(tot[:use])
            carry[s] = tot[use:]
        out.append((row, alive))
    return out


def write_files(tmp_path, lengths, seed=0):
    rng = np.random.default_rng(seed)
    datas, paths = [], []
    for i, n in enumerate(lengths):
        d = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
        p = tmp_path / f"s{i}.cf32"
        d.tofile(p)
        datas.append(d); paths.append(p)
    return datas, paths


@pytest.mark.parametrize("loop", [False, True])
@pytest.mark.parametrize("chunk,granule,lengths", [
    (256, 64, [1000, 256 * 3, 100, 5000]),      # ragged tail / exact multiple (one empty read before the rewind) / shorter than a chunk
    (4096, 64, [10000, 4096, 12288, 64]),
    (300, 16, [999, 1200, 300, 17]),            # chunk not a multiple of the granule: remainders are carried
])
def test_rounds_match_reference_semantics(tmp_path, loop, chunk, granule, lengths):
    import habdec_amd
    datas, paths = write_files(tmp_path, lengths)
    src = habdec_amd.IqFiles(paths, chunk=chunk, granule=granule, loop=loop)
    assert [src.count(s) for s in range(len(lengths))] == lengths
    for (row, alive), _ in zip(ref_rounds(datas, loop, chunk, granule, 40), range(40)):
        slab, n, got_alive = src.next()
        assert got_alive == alive
        for s, want in enumerate(row):
            assert n[s] == len(want) and n[s] % granule == 0
            assert np.array_equal(slab[s, :n[s]].view(np.uint32), want.view(np.uint32))
    if loop:
        assert all(src.rewinds(s) > 0 for s in range(len(lengths)))


def test_realtime_throttle_paces_the_rounds_like_the_reference(tmp_path):
    """IQSource_File.h:165-168 sleeps size_t(read / rate * 1000) milliseconds after a read when it plays a file in real time; the batch reader does
    that once per round for the longest read of the round.  Eight rounds of 4096 samples at 40960 samples/s are eight sleeps of 100 ms; the
    data is what the unthrottled reader delivers."""
    import time
    import habdec_amd
    datas, paths = write_files(tmp_path, [4096 * 8, 4096 * 8])
    fast = habdec_amd.IqFiles(paths, chunk=4096, granule=64)
    slow = habdec_amd.IqFiles(paths, chunk=4096, granule=64, realtime_rate=40960.0)
    t0 = time.perf_counter()
    rounds = [fast.next() for _ in range(8)]
    t_fast = time.perf_counter() - t0
    t0 = time.perf_counter()
    for want in rounds:
        slab, n, alive = slow.next()
        assert alive == want[2] and list(n) == list(want[1])
        assert np.array_equal(slab.view(np.uint32), want[0].view(np.uint32))
    t_slow = time.perf_counter() - t0
    # (lower bounds are the contract -- sleep_for never returns early; the upper bounds only say "the throttle is the sleep, not a busy loop gone wrong",
    # and are generous: a loaded CI host stretches every sleep)
    assert t_slow >= 0.8, t_slow                                    # 8 x 100 ms
    assert t_slow >= t_fast + 0.7 and t_slow < 30.0, (t_fast, t_slow)
    # the truncation to whole milliseconds (size_t(...)): 100 samples at 40960 samples/s are 2.44 ms -> 2 ms, not 0 and not 3
    tiny = habdec_amd.IqFiles(paths, chunk=128, granule=64, realtime_rate=40960.0)
    t0 = time.perf_counter()
    for _ in range(20):
        tiny.next()
    dt = time.perf_counter() - t0
    assert 0.06 <= dt < 30.0, dt                                    # 20 x 3 ms (128 samples: 3.125 ms -> 3 ms); no tight upper bound (loaded hosts)


def test_open_errors(tmp_path):
    import habdec_amd
    _, paths = write_files(tmp_path, [100])
    with pytest.raises(habdec_amd.HabdecError):
        habdec_amd.IqFiles([tmp_path / "missing.cf32"], chunk=64, granule=64)
    with pytest.raises(habdec_amd.HabdecError):
        habdec_amd.IqFiles(paths, chunk=32, granule=64)          # a round could never deliver a whole granule


@pytest.mark.gpu
@pytest.mark.parametrize("pipeline", [False, True])
def test_ingest_run_decodes_like_direct_feeding(tmp_path, pipeline):
    import habdec_amd
    from habdec_amd import synth
    fs, D, C, S = 2.048e6, 64, 65536, 6
    iq = []
    for s in range(S):
        text = synth.make_sentence(f"FILE{s}", f"{s},52.1,21.{s}")
        x = synth.fsk_iq(synth.rtty_bits(text * 2, 8, 2, 4, 4), fs, 300, seed=s, sigma=0.05)
        iq.append(x[:len(x) - len(x) % 1000 + 1000 * (s % 3)])       # ragged lengths, not multiples of D
    paths = []
    for s, x in enumerate(iq):
        p = tmp_path / f"f{s}.cf32"
        x.astype(np.complex64).tofile(p); paths.append(p)
    ref = habdec_amd.Engine(n_streams=S, max_chunk=C, sampling_rate=fs, decimation=D)
    files = [RefFile(x.astype(np.complex64), False) for x in iq]
    carry = [np.zeros(0, np.complex64)] * S
    total = 0
    while True:                                                     # the same rounds, fed through hd_process_host by hand
        slab = np.zeros((S, C), np.complex64); n = np.zeros(S, np.uint32); alive = 0
        for s in range(S):
            got = files[s].get(C - len(carry[s])); alive += len(got) > 0
            tot = np.concatenate([carry[s], got]); use = len(tot) - len(tot) % D
            slab[s, :use] = tot[:use]; n[s] = use; carry[s] = tot[use:]
        if not alive:
            break
        import ctypes as Ct
        habdec_amd.capi.check(ref.L.hd_process_host(ref.h, slab.ctypes.data, C, n.ctypes.data_as(Ct.POINTER(Ct.c_uint32)), 0))
        total += int(n.sum())
    want = [ref.take_sentences(s) for s in range(S)]
    assert sum(len(w) for w in want) >= S                            # the signals do decode

    eng = habdec_amd.Engine(n_streams=S, max_chunk=C, sampling_rate=fs, decimation=D, pipeline=pipeline)
    src = habdec_amd.IqFiles(paths, chunk=C, granule=D)
    assert eng.ingest(src) == total
    assert [eng.take_sentences(s) for s in range(S)] == want
    assert eng.rtty(0) == ref.rtty(0)


@pytest.mark.gpu
def test_ingest_sentences_equal_the_oracle_fed_like_the_reference_feeds_its_decoder(tmp_path):
    """The batched ingest against the ORACLE (not against the same engine fed by hand): every file read in 65536-sample requests like
    IQSource_File::get, pushed and processed per round like DECODER_THREAD does (main.cpp:234-245); the engine's sentences, characters
    and text must equal the oracle's stream by stream.  One file ends in a tail shorter than the first stage's history (the engine
    holds that back instead of failing the batch; the tail is idle carrier, so the text is unaffected)."""
    import habdec_amd
    from habdec_amd import synth
    from oracle import pyoracle
    fs, D, C, S = 2.048e6, 64, 65536, 5
    iq, paths = [], []
    for s in range(S):
        text = synth.make_sentence(f"ING{s}", f"{s},52.{s},21.{s}") * 2
        x = synth.fsk_iq(synth.rtty_bits(text, 8, 2, 4 + s, 6), fs, 300, seed=40 + s, sigma=0.06).astype(np.complex64)
        x = x[:(len(x) // C) * C + (100 if s == 1 else 3000 * s + 5000)]      # file 1 ends with a 100-sample read
        iq.append(x)
        p = tmp_path / f"g{s}.cf32"
        x.tofile(p); paths.append(p)
    eng = habdec_amd.Engine(n_streams=S, max_chunk=C, sampling_rate=fs, decimation=D, pipeline=True)
    done = eng.ingest(habdec_amd.IqFiles(paths, chunk=C, granule=D))
    assert done > 0
    for s in range(S):
        o = pyoracle.Decoder("oracle", factor=D)
        f = RefFile(iq[s], False)
        while True:
            got = f.get(C)
            if not len(got):
                break
            if len(got) >= 2176:                                    # (pushes shorter than the stage histories are undefined behaviour in the
                o(got, fs)                                          #  reference: the engine holds them back, the oracle is not given them)
        assert eng.take_sentences(s) == o.sentences(), s
        assert len(o.sentences()) == 2
        assert eng.take_chars(s) == o.text("chars_log"), s
    eng.close()


@pytest.mark.gpu
def test_ingest_keeps_looping_files_going_and_resumes_after_max_rounds(tmp_path):
    """Looping files of equal length, a whole number of chunks long: the reference's source returns one empty read before every rewind
    (IQSource_File.h:141-155), so every stream is empty in the same round -- that must not end a looping ingest; and a run bounded
    by max_rounds must leave the files where a second run continues seamlessly (nothing read ahead and dropped)."""
    import habdec_amd
    from habdec_amd import synth
    fs, D, C, S = 2.048e6, 64, 65536, 3
    paths, n_chunks = [], 3
    for s in range(S):
        x = synth.fsk_iq(np.ones(8, np.uint8), fs, 300, seed=70 + s, sigma=0.05, n_samples=n_chunks * C).astype(np.complex64)
        p = tmp_path / f"l{s}.cf32"
        x.tofile(p); paths.append(p)
    src = habdec_amd.IqFiles(paths, chunk=C, granule=D, loop=True)
    eng = habdec_amd.Engine(n_streams=S, max_chunk=C, sampling_rate=fs, decimation=D, pipeline=True)
    # rounds 0-2 deliver the file, round 3 is the empty read, rounds 4-6 the file again, round 7 empty, ...
    assert eng.ingest(src, max_rounds=5) == S * C * 4
    assert eng.ingest(src, max_rounds=6) == S * C * 5                # chunks 1, 2, the empty read, chunks 0, 1, 2: nothing was read ahead and dropped in between
    assert all(src.rewinds(s) == 2 for s in range(S))
    eng.close()
