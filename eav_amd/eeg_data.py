"""DataLoadEEG on MI355X - the reference's EEG pre-processing class (Dataload_eeg.py:35-160), same
constructor, methods and outputs, with the two data-heavy stages on the GPU in float64:

    downsampling()      scipy.signal.resample_poly(x, 1, fs_orig/fs_target)  -> eav_decimate_fir_f64
    bandpass_filter()   butter(5, band, 'bandpass', output='sos') + sosfilt    -> eav_sosfilt_f64

Filter *design* (101-tap Kaiser-windowed sinc; 5th-order Butterworth sections) stays on the host - it is a
few hundred flops - and follows scipy's published formulas (firwin / resample_poly, butter via scipy itself,
which is the reference's own dependency, requirements.txt).  `prepare_data()` returns what the reference
returns: (seg_f_div float64 [N, ch, 500], label_div int64 [N]); labels keep the reference's values
(argmax over the 10 one-hot rows, i.e. 1,3,5,7,9 - SURVEY Q8); `remap_labels=True` maps them to 0..4.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import _lib

SELECTED_CLASSES = [1, 3, 5, 7, 9]


def read_subject_recording(parent_directory, subject):
    """(signal [channel, time, trial], one-hot labels [10, trial]) of one subject, or None when the signal file is
    absent.  Host file I/O, outside the accelerated path."""
    from scipy.io import loadmat
    stem = f'subject{int(subject):02d}'
    folder = os.path.join(parent_directory, stem, 'EEG')
    signal_path, label_path = (os.path.join(folder, stem + suffix) for suffix in ('_eeg.mat', '_eeg_label.mat'))
    if not os.path.isfile(signal_path):
        return None
    contents = loadmat(signal_path)
    recording = next((np.asarray(contents[name]) for name in ('seg1', 'seg') if name in contents), None)
    if recording is None:
        raise KeyError(f"{signal_path}: neither 'seg1' nor 'seg' present")
    labels = np.asarray(loadmat(label_path)['label'])
    return np.swapaxes(recording, 0, 1), labels          # stored time-major -> channel-major


def resample_poly_design(up, down):
    """(h float64, center, n_out(n_in)) of scipy.signal.resample_poly(x, up, down) for up == 1:
    y[m] = sum_j h[j] x[m*down + center - j]  (window ('kaiser', 5.0), half_len = 10*max(up,down))."""
    if up != 1:
        raise NotImplementedError("only pure decimation (up == 1) is implemented")
    max_rate = max(up, down)
    half_len = 10 * max_rate
    m = np.arange(-half_len, half_len + 1, dtype=np.float64)
    fc = 1.0 / max_rate
    h = fc * np.sinc(fc * m) * np.kaiser(2 * half_len + 1, 5.0)      # scipy.signal.firwin(..., window=('kaiser',5))
    h /= h.sum()
    return h * up, half_len


def sos_tables(sos, Lc):
    """H [Lc, 2 nsec] (cascade output at step k from unit initial state j, zero input) and
    AL [2 nsec, 2 nsec] (state after Lc zero-input steps), float64 - simulated with scipy's own recurrence."""
    sos = np.asarray(sos, np.float64)
    nsec = sos.shape[0]
    ns = 2 * nsec
    z = np.zeros((ns, nsec, 2))
    for j in range(ns):
        z[j, j // 2, j % 2] = 1.0
    H = np.zeros((Lc, ns))
    for k in range(Lc):
        v = np.zeros(ns)
        for s in range(nsec):
            b0, b1, b2, _, a1, a2 = sos[s]
            o = b0 * v + z[:, s, 0]
            z[:, s, 0] = b1 * v - a1 * o + z[:, s, 1]
            z[:, s, 1] = b2 * v - a2 * o
            v = o
        H[k] = v
    AL = z.reshape(ns, ns).T.copy()            # column j = final state started from e_j
    return H, AL


def decimate(x_dev, down):
    """x_dev float64 device [nch, n] -> [nch, ceil(n/down)] (resample_poly(x, 1, down, axis=1))."""
    h, center = resample_poly_design(1, down)
    nch, n = x_dev.shape
    n_out = n // down + bool(n % down)
    hd = torch.from_numpy(h).to(x_dev.device)
    y = torch.empty(nch, n_out, dtype=torch.float64, device=x_dev.device)
    _lib.call("eav_decimate_fir_f64", x_dev.data_ptr(), hd.data_ptr(), y.data_ptr(), nch, n, n_out, down, len(h), center,
              _lib.stream_ptr())
    torch.cuda.current_stream().synchronize()
    return y


def sosfilt(sos, x_dev, chunk=2048):
    """scipy.signal.sosfilt(sos, x, axis=-1) for x float64 device [nch, n]."""
    sos = np.ascontiguousarray(sos, np.float64)
    nsec = sos.shape[0]
    nch, n = x_dev.shape
    Lc = int(chunk)
    H, AL = sos_tables(sos, Lc)
    dev = x_dev.device
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    sd, Hd, ALd = t(sos), t(H), t(AL)
    nchunk = (n + Lc - 1) // Lc
    zend = torch.empty(nch, nchunk, 2 * nsec, dtype=torch.float64, device=dev)
    zstart = torch.empty_like(zend)
    y = torch.empty_like(x_dev)
    _lib.call("eav_sosfilt_f64", x_dev.data_ptr(), y.data_ptr(), sd.data_ptr(), Hd.data_ptr(), ALd.data_ptr(),
              zend.data_ptr(), zstart.data_ptr(), nch, n, nsec, Lc, _lib.stream_ptr())
    torch.cuda.current_stream().synchronize()
    return y


class DataLoadEEG:
    def __init__(self, subject=1, band=[0.3, 50], fs_orig=500, fs_target=100, parent_directory='./Datasets/EAV',
                 device=None, remap_labels=False):
        self.subject = subject
        self.band = band
        self.fs_orig = fs_orig
        self.fs_target = fs_target
        self.parent_directory = parent_directory
        self.device = torch.device(device if device else ("cuda" if torch.cuda.is_available() else "cpu"))
        self.remap_labels = remap_labels
        self.seg = None
        self.label = None
        self.seg_f = None
        self.seg_f_div = None
        self.label_div = None

    def _dev(self, a):
        if self.device.type != "cuda":
            raise _lib.EavError("eav_amd.DataLoadEEG runs its filters on an MI355X (no CPU fallback)")
        return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64).to(self.device)

    def load_mat_data(self):
        """Reads one subject's recording from disk (the file layout Dataload_eeg.py:54-83 expects:
        <parent>/subjectNN/EEG/subjectNN_eeg.mat with variable `seg1` or `seg` [time, channel, trial] and
        subjectNN_eeg_label.mat with `label` [10, trials]); leaves self.seg as [channel, time, trial]."""
        found = read_subject_recording(self.parent_directory, self.subject)
        tag = f'subject{self.subject:02d}'
        if found is None:
            print(f'[Error] EEG data not found for {tag}')
            return
        self.seg, self.label = found
        print(f'[Info] Loaded EEG data for {tag}')

    def downsampling(self):
        # Dataload_eeg.py:85-102
        if self.seg is None:
            return
        ch, t, tri = self.seg.shape
        factor = self.fs_target / self.fs_orig
        down_factor = int(self.fs_orig / self.fs_target)
        seg = self._dev(self.seg)
        tm = seg.permute(0, 2, 1).reshape(ch, tri * t).contiguous()       # == np.reshape(seg, [ch, t*tri], order='F')
        tm2 = decimate(tm, down_factor)
        new_time = int(t * factor)
        self.seg = tm2.reshape(ch, tri, new_time).permute(0, 2, 1).contiguous()   # order='F' reshape back

    def bandpass_filter(self):
        # Dataload_eeg.py:104-121
        if self.seg is None:
            return
        from scipy.signal import butter
        seg = self.seg if isinstance(self.seg, torch.Tensor) else self._dev(self.seg)
        ch, t, tri = seg.shape
        dat = seg.permute(0, 2, 1).reshape(ch, tri * t).contiguous()
        sos = butter(5, self.band, btype='bandpass', fs=self.fs_target, output='sos')
        fdat = sosfilt(sos, dat)
        self.seg_f = fdat.reshape(ch, tri, t).permute(0, 2, 1).contiguous()

    def segment_and_select_classes(self):
        # Dataload_eeg.py:123-152 (20 s trials -> 4 x 5 s windows, listening classes only)
        if self.seg_f is None:
            return
        ch, t, tri = self.seg_f.shape
        win = 500
        nwin = t // win
        # tm1[c, a, b, d] = seg_f[c, a + win*b, d]; flattened F-order over (b, d): index b + nwin*d
        tm1 = self.seg_f[:, :win * nwin, :].reshape(ch, nwin, win, tri).permute(0, 2, 1, 3)     # [c, a, b, d]
        seg_div = tm1.permute(0, 1, 3, 2).reshape(ch, win, tri * nwin)                           # [c, a, d*nwin + b]
        label_div = np.repeat(self.label, repeats=nwin, axis=1)
        cls = np.argmax(label_div, axis=0)
        mask = np.isin(cls, SELECTED_CLASSES)
        idx = torch.from_numpy(np.flatnonzero(mask)).to(seg_div.device)
        data_subset = seg_div.index_select(2, idx)
        self.seg_f_div = data_subset.permute(2, 0, 1).contiguous().cpu().numpy()
        lab = np.argmax(label_div[:, mask], axis=0)
        self.label_div = (lab - 1) // 2 if self.remap_labels else lab

    def prepare_data(self):
        self.load_mat_data()
        self.downsampling()
        self.bandpass_filter()
        self.segment_and_select_classes()
        return self.seg_f_div, self.label_div

    data_prepare = prepare_data      # the name the EEGNet driver calls (EEGNet_tor.py:149)
