"""The index arithmetic of csrc/eegnet_fir_fft.hip, restated in numpy and checked on the CPU: the in-wave 1024-point FFT
(radix 16 x 16 x 4, 16 points per lane, two exchanges through LDS slots), the bank-conflict freedom of both exchanges under
the MI355X lane-group rules (ds_write_b64: groups of 16 lanes over 32 banks of 4 B, ds_read_b64: groups of 32 lanes over 64
banks), and the two frequency-domain identities the kernels rest on (overlap-save correlation for the forward; the
conjugate-product accumulation for the weight gradient, with two electrodes packed into one complex signal).
Reference ops: nn.Conv2d(1, 8, (1, 300), padding='same') and its weight gradient - CNN_torch/EEGNet_tor.py:24,51,109."""
import numpy as np

N, LB, K = 1024, 704, 300
P1, P2 = 68, 260          # slot pitches of the two exchanges (eegnet_fir_fft.hip: fft1024)


def W(m):
    return np.exp(-2j * np.pi * (m % N) / N)


def dft4(a, inv):
    a0, a1, a2, a3 = a
    t0, t1, t2, t3 = a0 + a2, a0 - a2, a1 + a3, (a1 - a3) * (1j if inv else -1j)
    return [t0 + t2, t1 + t3, t0 - t2, t1 - t3]


def dft16(v, inv):
    y = [[None] * 4 for _ in range(4)]
    for b in range(4):
        r = dft4([v[b], v[4 + b], v[8 + b], v[12 + b]], inv)
        for c in range(4):
            w = np.exp(-2j * np.pi * (b * c) / 16)
            y[b][c] = r[c] * (np.conj(w) if inv else w)
    X = [None] * 16
    for c in range(4):
        r = dft4([y[0][c], y[1][c], y[2][c], y[3][c]], inv)
        for d in range(4):
            X[c + 4 * d] = r[d]
    return X


def fft1024(x, inv=False):
    """Lane l holds x[l + 64 j] in register j on input and X[l + 64 j] on output - the kernel's layout."""
    regs = [[x[lane + 64 * j] for j in range(16)] for lane in range(64)]
    lds = np.zeros(1088, complex)
    for lane in range(64):
        regs[lane] = dft16(regs[lane], inv)
        for k1 in range(16):
            t = W(lane * k1)
            regs[lane][k1] *= np.conj(t) if inv else t
    for lane in range(64):
        for k1 in range(16):
            lds[P1 * k1 + lane] = regs[lane][k1]
    for lane in range(64):
        regs[lane] = [lds[P1 * (lane >> 2) + 4 * n2 + (lane & 3)] for n2 in range(16)]
    for lane in range(64):
        regs[lane] = dft16(regs[lane], inv)
        for k2 in range(16):
            t = W(16 * (lane & 3) * k2)
            regs[lane][k2] *= np.conj(t) if inv else t
    for lane in range(64):
        for k2 in range(16):
            lds[(lane >> 2) + 16 * k2 + P2 * (lane & 3)] = regs[lane][k2]
    X = np.zeros(N, complex)
    for lane in range(64):
        for m in range(4):
            r = dft4([lds[lane + 64 * m + P2 * n3] for n3 in range(4)], inv)
            for k3 in range(4):
                X[lane + 64 * (m + 4 * k3)] = r[k3]
    return X


def test_in_wave_fft_matches_numpy():
    rng = np.random.default_rng(0)
    x = rng.normal(size=N) + 1j * rng.normal(size=N)
    assert np.abs(fft1024(x) - np.fft.fft(x)).max() < 1e-11
    assert np.abs(fft1024(x, True) / N - np.fft.ifft(x)).max() < 1e-13


def _ways(slots, group, banks_in_slots):
    worst = 1
    for g in range(0, 64, group):
        s = [a % banks_in_slots for a in slots[g:g + group]]
        worst = max(worst, max(s.count(v) for v in set(s)))
    return worst


def test_both_exchanges_are_bank_conflict_free():
    # 8-byte slots: a ds_write_b64 group of 16 lanes covers 32 banks = 16 slots, a ds_read_b64 group of 32 lanes 64 banks = 32 slots
    for k in range(16):
        assert _ways([P1 * k + lane for lane in range(64)], 16, 16) == 1
        assert _ways([P1 * (lane >> 2) + 4 * k + (lane & 3) for lane in range(64)], 32, 32) == 1
        assert _ways([(lane >> 2) + 16 * k + P2 * (lane & 3) for lane in range(64)], 16, 16) == 1
    for m in range(4):
        for n3 in range(4):
            assert _ways([lane + 64 * m + P2 * n3 for lane in range(64)], 32, 32) == 1
    assert max(P1 * 15 + 63, 15 + 16 * 15 + P2 * 3) < 1088          # both images fit the wave's exchange buffer
    # (the first choice of the second pitch, 264, is 2-way conflicted on the writes: the reason for 260)
    assert max(_ways([(lane >> 2) + 16 * k + 264 * (lane & 3) for lane in range(64)], 16, 16) for k in range(16)) == 2


def test_overlap_save_identities():
    rng = np.random.default_rng(1)
    w = rng.normal(size=K)
    s = rng.normal(size=N) + 1j * rng.normal(size=N)                 # two electrodes packed: real + i imag
    H = np.conj(np.fft.fft(np.r_[w, np.zeros(N - K)])) / N
    y = fft1024(fft1024(s) * H, True)
    ref = np.array([np.dot(w, s[j:j + K]) for j in range(LB)])       # y[j] = sum_k w[k] s[j + k], no wrap for j < 704
    assert np.abs(y[:LB] - ref).max() < 1e-11
    assert LB + K - 1 <= N and N - LB + 1 == 321                     # the longest kernel the block length admits
    d = np.zeros(N, complex)
    d[:LB] = rng.normal(size=LB) + 1j * rng.normal(size=LB)          # dy of the two electrodes, zero-padded
    R = fft1024(fft1024(s) * np.conj(fft1024(d)), True) / N
    ref = np.array([np.dot(d[:LB].real, s[k:k + LB].real) + np.dot(d[:LB].imag, s[k:k + LB].imag) for k in range(K)])
    assert np.abs(R[:K].real - ref).max() < 1e-10                    # Re(.) = sum over BOTH electrodes of the pair
