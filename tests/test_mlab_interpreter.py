"""oracle/mlab (the interpreter that executes the reference's .m files into tests/golden/ref_*) against MATLAB's DOCUMENTED
behaviour: each case is a statement list whose value MathWorks' documentation (or the language definition) fixes - the colon
operator's element count and end point, rem / mod signs, round half away from zero, var's N-1, max's first occurrence, sort's
stability, column-major linear indexing, `end` arithmetic, auto-growth, struct arrays, value semantics, switch on strings,
anonymous-function capture, fread / fseek / ftell.  These are the rules the reference's hot path leans on; a
misreading of one of them would be common to the fixtures and the oracle, so they are pinned here one by one."""
import os

import numpy as np
import pytest

from oracle import mlab

_N = [0]


def run(tmp_path, body, nargout=1, files=None):
    """Executes `body` as the body of a function with outputs r (r1, r2, ... for nargout > 1) and returns them."""
    _N[0] += 1
    name = f"case{_N[0]}"
    outs = "r" if nargout == 1 else "[" + ", ".join(f"r{i + 1}" for i in range(nargout)) + "]"
    with open(os.path.join(tmp_path, name + ".m"), "w") as f:
        f.write(f"function {outs} = {name}()\n{body}\nend\n")
    for fn, text in (files or {}).items():
        with open(os.path.join(tmp_path, fn), "w") as f:
            f.write(text)
    I = mlab.Interpreter([str(tmp_path)])
    r = I.call(name, nargout=nargout)
    conv = lambda v: mlab.from_matlab(v)  # noqa: E731
    return conv(r) if nargout == 1 else tuple(conv(v) for v in r)


def vec(x):
    return np.asarray(x, dtype=np.float64).reshape(-1)


def test_colon_operator(tmp_path):
    r = vec(run(tmp_path, "r = 0:0.1:1;"))
    assert np.allclose(r, np.linspace(0, 1, 11), atol=2e-16)
    assert r.size == 11 and r[-1] == 1.0 and r[0] == 0.0                 # the end point is hit exactly
    assert vec(run(tmp_path, "r = 0:0.1:0.3;")).size == 4                 # 3*0.1 > 0.3 in floating point, still 4 elements
    assert vec(run(tmp_path, "r = numel(1:0);"))[0] == 0                  # empty range
    assert np.array_equal(vec(run(tmp_path, "r = 10:-3:0;")), [10, 7, 4, 1])
    assert np.array_equal(vec(run(tmp_path, "r = size(zeros(1, 0));")), [1, 0])
    assert np.array_equal(vec(run(tmp_path, "r = (1:3) * 2;")), [2, 4, 6])
    # symmetric construction: the middle of a long ramp is computed from both ends (what tracking.m's code ramps rely on)
    r = vec(run(tmp_path, "r = 0.1:0.05683:1023.1;"))
    n = int(np.floor((1023.1 - 0.1) / 0.05683 * (1 + 2 ** -52))) + 1
    assert r.size == n and r[0] == 0.1 and abs(r[-1] - (0.1 + (n - 1) * 0.05683)) < 1e-12


def test_rounding_and_remainders(tmp_path):
    assert np.array_equal(vec(run(tmp_path, "r = round([2.5 -2.5 0.5 -0.5 1.4999]);")), [3, -3, 1, -1, 1])   # half away from zero
    assert np.array_equal(vec(run(tmp_path, "r = [rem(-7, 3) mod(-7, 3) rem(7, -3) mod(7, -3) rem(5, 0.5)];")), [-1, 2, 1, -2, 0])
    assert np.array_equal(vec(run(tmp_path, "r = [fix(-2.7) floor(-2.7) ceil(-2.7) ceil(2) ceil(2.0000001)];")), [-2, -3, -2, 2, 3])
    r = vec(run(tmp_path, "r = rem(-7.5, 2*pi);"))
    assert r[0] < 0 and abs(r[0] - np.fmod(-7.5, 2 * np.pi)) < 1e-15      # sign of the dividend
    assert np.array_equal(vec(run(tmp_path, "r = [7/2 floor(7/2) 2^10 mod(10, 3)];")), [3.5, 3, 1024, 1])


def test_statistics(tmp_path):
    assert abs(vec(run(tmp_path, "r = var([1 2 3 4]);"))[0] - 5.0 / 3.0) < 1e-15          # N - 1
    assert abs(vec(run(tmp_path, "r = var([1+2i, 3-1i, -2+0.5i]);"))[0] - np.var([1 + 2j, 3 - 1j, -2 + 0.5j], ddof=1)) < 1e-15
    assert abs(vec(run(tmp_path, "r = std([2 4 4 4 5 5 7 9]);"))[0] - np.std([2, 4, 4, 4, 5, 5, 7, 9], ddof=1)) < 1e-15
    assert np.array_equal(vec(run(tmp_path, "r = mean([1 2; 3 4]);")), [2, 3])             # down the columns
    assert np.array_equal(vec(run(tmp_path, "r = sum([1 2; 3 4], 2);")), [3, 7])
    assert np.array_equal(vec(run(tmp_path, "r = cumsum([1 2 3 4]);")), [1, 3, 6, 10])


def test_max_min_sort_find(tmp_path):
    m, i = run(tmp_path, "[r1, r2] = max([3 9 9 2]);", nargout=2)
    assert vec(m)[0] == 9 and vec(i)[0] == 2                                                 # first occurrence
    m, i = run(tmp_path, "A = [1 8 3; 7 2 8];\n[r1, r2] = max(max(A, [], 2));", nargout=2)
    assert vec(m)[0] == 8 and vec(i)[0] == 1                                                 # max over rows, then the first row holding it
    m, i = run(tmp_path, "A = [1 8 3; 7 2 8];\n[r1, r2] = max(max(A));", nargout=2)
    assert vec(m)[0] == 8 and vec(i)[0] == 2                                                 # column maxima [7 8 8]: first 8 at column 2
    v, i = run(tmp_path, "[r1, r2] = sort([3 1 3 2 1], 'descend');", nargout=2)
    assert np.array_equal(vec(v), [3, 3, 2, 1, 1]) and np.array_equal(vec(i), [1, 3, 4, 2, 5])   # stable
    assert np.array_equal(vec(run(tmp_path, "r = find([0 3 0 5] > 1);")), [2, 4])
    assert np.array_equal(vec(run(tmp_path, "[~, r] = min([4 2 2 9]);")), [2])
    assert np.array_equal(vec(run(tmp_path, "r = [any([0 0 1]) all([1 1 0]) isempty([]) numel(zeros(3, 4)) length(zeros(3, 7))];")), [1, 0, 1, 12, 7])


def test_indexing_growth_and_value_semantics(tmp_path):
    assert np.array_equal(vec(run(tmp_path, "A = [1 2 3; 4 5 6];\nr = [A(2) A(3) A(end) A(2, end) A(end, 1)];")), [4, 2, 6, 6, 4])   # column-major
    assert np.array_equal(vec(run(tmp_path, "A = [1 2 3; 4 5 6];\nr = A(:).';")), [1, 4, 2, 5, 3, 6])
    assert np.array_equal(vec(run(tmp_path, "a = [1 2 3];\na(end+1) = 9;\na(7) = 1;\nr = a;")), [1, 2, 3, 9, 0, 0, 1])                 # auto-growth pads with zeros
    assert np.array_equal(vec(run(tmp_path, "a = 1:6;\na([2 4]) = [];\nr = a;")), [1, 3, 5, 6])
    assert np.array_equal(vec(run(tmp_path, "a = 1:5;\nr = a(a > 2 & a < 5);")), [3, 4])
    assert np.array_equal(vec(run(tmp_path, "a = [5 6 7];\nb = a;\nb(2) = 0;\nr = a;")), [5, 6, 7])                                    # assignment copies
    assert np.array_equal(vec(run(tmp_path, "a = 10:10:50;\nr = a(end-1:end);")), [40, 50])
    assert np.array_equal(vec(run(tmp_path, "c = [1 2 3];\nr = [c(end) c c(1)];")), [3, 1, 2, 3, 1])                                    # tracking.m:158's padding
    assert np.array_equal(vec(run(tmp_path, "x = zeros(1, 4);\nx(2:3) = [7 8];\nr = x;")), [0, 7, 8, 0])
    assert np.array_equal(vec(run(tmp_path, "r = reshape(1:6, 2, 3);\nr = r(2, :);")), [2, 4, 6])
    assert np.array_equal(vec(run(tmp_path, "r = repmat([1 2], 1, 3);")), [1, 2, 1, 2, 1, 2])
    assert np.array_equal(vec(run(tmp_path, "r = circshift([1 2 3 4], 1);")), [4, 1, 2, 3])
    assert np.array_equal(vec(run(tmp_path, "r = fliplr([1 2 3]);")), [3, 2, 1])


def test_complex_and_transposes(tmp_path):
    r = np.asarray(run(tmp_path, "z = [1+2i 3-4i];\nr = [z' ; z.'];")).reshape(-1)
    assert np.array_equal(r, [1 - 2j, 3 + 4j, 1 + 2j, 3 - 4j])                                 # ' conjugates, .' does not
    assert abs(np.asarray(run(tmp_path, "r = exp(1i*pi);")).reshape(-1)[0] + 1) < 1e-15
    assert np.array_equal(vec(run(tmp_path, "r = abs([3+4i, -5]);")), [5, 5])
    x = np.arange(8) + 1j * np.arange(8)[::-1]
    got = np.asarray(run(tmp_path, "x = (0:7) + 1i*(7:-1:0);\nr = ifft(fft(x) .* conj(fft(x)));")).reshape(-1)
    assert np.allclose(got, np.fft.ifft(np.fft.fft(x) * np.conj(np.fft.fft(x))), atol=1e-12)
    assert np.array_equal(vec(run(tmp_path, "r = real([1+2i 3]) + imag([1+2i 3]);")), [3, 3])
    assert abs(vec(run(tmp_path, "r = atan(1/0);"))[0] - np.pi / 2) < 1e-16 and np.isnan(vec(run(tmp_path, "r = atan(0/0);"))[0])   # the loop's atan(Q/I) with I = 0


def test_structs_cells_strings_and_control_flow(tmp_path):
    r = run(tmp_path, "s.a = 1;\ns(3).a = 7;\nr = [numel(s) isempty(s(2).a) s(3).a];")
    assert np.array_equal(vec(r), [3, 1, 7])                                                     # struct arrays grow, new elements hold []
    r = run(tmp_path, "t.x = zeros(1, 3);\nt = repmat(t, 1, 2);\nt(2).x(2) = 5;\nr = [t(1).x t(2).x];")
    assert np.array_equal(vec(r), [0, 0, 0, 0, 5, 0])
    r = run(tmp_path, "s.f = 2;\nname = 'f';\ns.(name) = s.(name) + 1;\nr = [s.f isfield(s, 'f') isfield(s, 'g')];")
    assert np.array_equal(vec(r), [3, 1, 0])
    r = run(tmp_path, "c = {1, 'ab', [4 5]};\nr = [numel(c) c{3}(2) numel(c{2})];")
    assert np.array_equal(vec(r), [3, 5, 2])
    r = run(tmp_path, "k = 0;\nswitch 'schar'\n case {'int8', 'schar'}\n  k = 1;\n case 'int16'\n  k = 2;\n otherwise\n  k = 3;\nend\nr = k;")
    assert vec(r)[0] == 1
    r = run(tmp_path, "k = 0;\nfor v = [1 2; 3 4]\n k = k * 10 + v(1) + v(2);\nend\nr = k;")
    assert vec(r)[0] == 46                                                                        # `for` walks the COLUMNS of a matrix
    r = run(tmp_path, "k = 0;\nfor i = 1:10\n if i == 3, continue; end\n if i > 5, break; end\n k = k + i;\nend\nr = k;")
    assert vec(r)[0] == 1 + 2 + 4 + 5
    r = run(tmp_path, "a = 5;\nf = @(x) x + a;\na = 100;\nr = f(1);")
    assert vec(r)[0] == 6                                                                         # anonymous functions capture at creation
    r = run(tmp_path, "r = [strcmp('abc', 'abc') strcmp('abc', 'abd') numel(['ab' 'cd']) double('A')];")
    assert np.array_equal(vec(r), [1, 0, 4, 65])
    r = run(tmp_path, "r = helper2(3);", files={"helper2.m": "function [a, b] = helper2(x)\na = x * 2;\nif nargout > 1, b = 1; end\nend\n"})
    assert vec(r)[0] == 6
    r = run(tmp_path, "x = 3;\nwhile x > 0\n x = x - 2;\nend\nr = x;")
    assert vec(r)[0] == -1
    assert np.array_equal(vec(run(tmp_path, "r = [1 -1];")), [1, -1]) and np.array_equal(vec(run(tmp_path, "a = 2;\nr = [a -1];")), [2, -1]) \
        and np.array_equal(vec(run(tmp_path, "a = 2;\nr = [a - 1];")), [1]) and np.array_equal(vec(run(tmp_path, "a = 2;\nr = [a-1];")), [1])   # whitespace rule in []


def test_file_reads_as_the_reference_does_them(tmp_path):
    """fread(fid, n, 'int8')' after fseek / ftell, as tracking.m:145-153,212-236 use them."""
    _N[0] += 1
    name = f"case{_N[0]}"
    with open(os.path.join(tmp_path, name + ".m"), "w") as f:
        f.write(f"function [r1, r2, r3, r4] = {name}(fid)\nfseek(fid, 4, 'bof');\nr1 = ftell(fid);\n[d, n] = fread(fid, 6, 'int8');\nr2 = d';\nr3 = n;\n"
                "[d, n] = fread(fid, 100, 'int8');\nr4 = n;\nend\n")
    I = mlab.Interpreter([str(tmp_path)])
    data = np.arange(-8, 8, dtype=np.int8)
    fid = mlab.register_file(I, data.tobytes(), "/data/x.bin")
    r1, r2, r3, r4 = I.call(name, fid, nargout=4)
    assert vec(mlab.from_matlab(r1))[0] == 4 and np.array_equal(vec(mlab.from_matlab(r2)), data[4:10]) and vec(mlab.from_matlab(r3))[0] == 6
    assert vec(mlab.from_matlab(r4))[0] == 6                                                       # a short read returns what is left


def test_xcorr_kron_height_and_logical_assignment_as_navdecoding_uses_them(tmp_path):
    """The built-ins the bit-synchronisation blocks of NAVdecoding.m lean on (tests/golden/ref_navsync_*), against MATLAB's
    documented behaviour: xcorr(x, y) of a row and a shorter row returns 2 * max(N, M) - 1 lags in a ROW, lag k at index N + k,
    c(N + k) = sum_n x(n + k) * y(n) with the shorter input zero-padded; kron of two rows; height / width; `v(v > 0) = 1;
    v(v <= 0) = -1` leaves zeros at -1; find(...)' of a row is a column (the loops run over height(index)); round half away."""
    r = vec(run(tmp_path, "x = [1 2 3 4 5]; y = [1 -1]; r = xcorr(x, y);"))
    assert r.size == 9
    full = [sum((x if 0 <= (n + k) < 5 else 0) * ([1, -1][n] if n < 2 else 0) for n in range(5) for x in [[1, 2, 3, 4, 5][n + k] if 0 <= n + k < 5 else 0]) for k in range(-4, 5)]
    assert np.array_equal(r, np.array(full, dtype=float))
    assert np.array_equal(r[4:], [1 - 2, 2 - 3, 3 - 4, 4 - 5, 5])         # the non-negative lags the sync blocks look at
    sz = vec(run(tmp_path, "a = xcorr([1 2 3], [1 1]); b = xcorr([1; 2; 3], [1; 1]); r = [size(a) size(b)];"))
    assert np.array_equal(sz, [1, 5, 5, 1])
    assert np.array_equal(vec(run(tmp_path, "r = kron([1 -1 1], [1 2]);")), [1, 2, -1, -2, 1, 2])
    assert np.array_equal(vec(run(tmp_path, "v = [3 0 -2 0.5]; v(v > 0) = 1; v(v <= 0) = -1; r = v;")), [1, -1, -1, 1])
    assert np.array_equal(vec(run(tmp_path, "idx = find([0 1 0 1 1] > 0)'; r = [height(idx) width(idx) size(idx)];")), [3, 1, 3, 1])
    assert np.array_equal(vec(run(tmp_path, "r = round([9.5 -9.5 9.49 239.99]);")), [10, -10, 9, 240])


def test_run_lines_executes_only_the_given_lines_of_a_file(tmp_path):
    """Interpreter.run_lines: a SECTION of a function file as a script in a given workspace (the NAVdecoding.m sync blocks) - the
    other lines are blanked, not removed: an error still names the file's own line."""
    path = os.path.join(tmp_path, "section.m")
    with open(path, "w") as f:
        f.write("function r = section(x)\nunknownToolboxCall();\na = x * 2;\nb = a + 1;\nunknownToolboxCall();\nc = b * undefinedName;\nend\n")
    I = mlab.Interpreter([str(tmp_path)])
    ws = I.run_lines(path, [(3, 4)], {"x": mlab.to_matlab(5.0)})
    assert float(mlab.from_matlab(ws["b"])) == 11.0 and "c" not in ws
    with pytest.raises(mlab.MError) as e:
        I.run_lines(path, [(3, 4), (6, 6)], {"x": mlab.to_matlab(5.0)})
    assert "section.m:6" in str(e.value)


# ---- numeric built-ins the fixtures and the oracle would otherwise share with ONE library routine -------------------------------
# oracle/mlab's fir1 / filtfilt / fft / xcorr call SciPy / NumPy, and so does oracle/gnss_oracle.py: a wrong reading of MATLAB's
# definition (or a library quirk) would be common to both sides of every fixture comparison.  Each is therefore checked here
# against MATLAB's DOCUMENTED definition written out independently - closed forms and plain loops, no signal-processing library.
def _sinc(x):
    x = np.asarray(x, dtype=np.float64)
    out = np.ones_like(x)
    nz = x != 0
    out[nz] = np.sin(np.pi * x[nz]) / (np.pi * x[nz])
    return out


def test_fir1_is_the_hamming_windowed_ideal_filter_scaled_at_the_passband_centre(tmp_path):
    """fir1(n, [w1 w2]) (acquisition.m:56-58 calls fir1(700, ...)): ideal band-pass impulse response w2 sinc(w2 k) - w1 sinc(w1 k),
    k = m - n/2, times hamming(n + 1) = 0.54 - 0.46 cos(2 pi m / n), scaled so that the magnitude response is exactly 1 at the
    centre of the pass band (w1 + w2) / 2 (frequencies normalised to Nyquist = 1); fir1(n, w): low-pass, unit gain at DC."""
    for order, w1, w2 in ((700, 0.2301, 0.7912), (50, 0.1, 0.35), (21, 0.4, 0.9)):
        got = vec(run(tmp_path, f"r = fir1({order}, [{w1!r} {w2!r}]);"))
        m = np.arange(order + 1, dtype=np.float64)
        k = m - order / 2.0
        h = (w2 * _sinc(w2 * k) - w1 * _sinc(w1 * k)) * (0.54 - 0.46 * np.cos(2.0 * np.pi * m / order))
        f0 = (w1 + w2) / 2.0
        h = h / abs(np.sum(h * np.exp(-1j * np.pi * f0 * m)))
        assert got.shape == h.shape and np.max(np.abs(got - h)) < 1e-14, (order, np.max(np.abs(got - h)))
    got = vec(run(tmp_path, "r = fir1(30, 0.25);"))
    m = np.arange(31, dtype=np.float64)
    h = 0.25 * _sinc(0.25 * (m - 15.0)) * (0.54 - 0.46 * np.cos(2.0 * np.pi * m / 30.0))
    assert np.max(np.abs(got - h / np.sum(h))) < 1e-15


def _filtfilt_by_hand(b, x):
    """MATLAB's filtfilt(b, 1, x), as documented: the signal extended at both ends by nfact = 3 (nfilt - 1) samples reflected
    about the end points (2 x(1) - x(nfact + 1 : -1 : 2) ...), filtered forwards with initial conditions that make the filter
    start in steady state for the first extended sample, reversed, filtered again the same way, reversed and trimmed.  For an
    FIR filter in transposed direct form II those initial conditions zi(k) = sum_{j >= k} b(j + 1), times the first sample, ARE a
    signal that was constant at its first value before it began - which is how the loops below start."""
    b = [float(v) for v in b]
    nfact = 3 * (len(b) - 1)
    ext = [2.0 * x[0] - x[i] for i in range(nfact, 0, -1)] + [float(v) for v in x] + [2.0 * x[-1] - x[-1 - i] for i in range(1, nfact + 1)]

    def fir_from_steady_state(sig):
        out = []
        for n_ in range(len(sig)):
            acc = 0.0
            for k_, bk in enumerate(b):
                acc += bk * (sig[n_ - k_] if n_ - k_ >= 0 else sig[0])
            out.append(acc)
        return out
    y = fir_from_steady_state(ext)
    y = fir_from_steady_state(y[::-1])[::-1]
    return np.array(y[nfact:len(y) - nfact])


def test_filtfilt_is_two_steady_state_fir_passes_over_the_odd_reflected_signal(tmp_path):
    rng = np.random.default_rng(5)
    for nb, nx in ((9, 120), (21, 200), (4, 40)):
        b = rng.standard_normal(nb)
        x = rng.standard_normal(nx) + 0.7           # a level: the start-up transient the initial conditions remove would show
        body = "b = [" + " ".join(repr(float(v)) for v in b) + "];\nx = [" + " ".join(repr(float(v)) for v in x) + "];\nr = filtfilt(b, 1, x);"
        got = vec(run(tmp_path, body))
        want = _filtfilt_by_hand(b, x)
        assert got.shape == want.shape and np.max(np.abs(got - want)) < 1e-11 * max(1.0, np.max(np.abs(want))), (nb, nx, np.max(np.abs(got - want)))
    # a complex signal (longSignal is complex, acquisition.m:60) is filtered component by component
    x = rng.standard_normal(64) + 1j * rng.standard_normal(64)
    b = rng.standard_normal(7)
    body = ("b = [" + " ".join(repr(float(v)) for v in b) + "];\nx = [" + " ".join(repr(float(v.real)) for v in x) + "] + 1i * [" +
            " ".join(repr(float(v.imag)) for v in x) + "];\nr = filtfilt(b, 1, x);")
    got = np.asarray(run(tmp_path, body)).reshape(-1)
    want = _filtfilt_by_hand(b, x.real) + 1j * _filtfilt_by_hand(b, x.imag)
    assert np.max(np.abs(got - want)) < 1e-11


def _dft(x, sign):
    """X[k] = sum_n x[n] exp(sign 2 pi i n k / N), term by term in float64; the angle is reduced with integer arithmetic
    (n k mod N) so that no large argument reaches the exponential."""
    n = x.shape[0]
    idx = (np.arange(n)[:, None] * np.arange(n)[None, :]) % n
    return (np.exp(sign * 2j * np.pi * idx / n) * x[None, :]).sum(axis=1)


@pytest.mark.parametrize("n", [240, 1250, 36])
def test_fft_and_ifft_are_the_textbook_sums_for_the_mixed_radix_sizes_the_searches_use(tmp_path, n):
    """Y = fft(X): Y(k) = sum_j X(j) W^((j-1)(k-1)), W = exp(-2 pi i / n); X = ifft(Y): (1/n) sum_k Y(k) W^(-(j-1)(k-1)).  The
    searches transform 36 000 = 2^5 3^2 5^3 (and 24 000, 144 000, ...) points: 240 = 2^4 3 5 and 1 250 = 2 5^4 run through the
    same radix-2/3/5 code paths of the library the interpreter and the oracle share, against an O(n^2) sum that shares nothing."""
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    lit = "[" + " ".join(repr(float(v.real)) for v in x) + "] + 1i * [" + " ".join(repr(float(v.imag)) for v in x) + "]"
    got = np.asarray(run(tmp_path, f"x = {lit};\nr = fft(x);")).reshape(-1)
    want = _dft(x, -1.0)
    assert np.max(np.abs(got - want)) < 1e-11 * np.max(np.abs(want))
    got = np.asarray(run(tmp_path, f"x = {lit};\nr = ifft(x);")).reshape(-1)
    assert np.max(np.abs(got - _dft(x, +1.0) / n)) < 1e-12
    # zero padding: fft([c zeros(1, n)]) is what acquisition.m:160-162 transforms
    got = np.asarray(run(tmp_path, f"x = {lit};\nr = fft([x zeros(1, {n})]);")).reshape(-1)
    want = _dft(np.concatenate([x, np.zeros(n)]), -1.0)
    assert np.max(np.abs(got - want)) < 1e-11 * np.max(np.abs(want))


def test_xcorr_is_the_lag_sum_of_the_longer_length(tmp_path):
    """c = xcorr(x, y): c(m + N) = sum_n x(n + m) conj(y(n)) for lags m = -(N - 1) .. N - 1, N = the longer length, the shorter
    vector zero-padded (NAVdecoding.m correlates the bit stream with the preamble this way)."""
    rng = np.random.default_rng(9)
    x, y = rng.standard_normal(37), rng.standard_normal(11)
    body = "x = [" + " ".join(repr(float(v)) for v in x) + "];\ny = [" + " ".join(repr(float(v)) for v in y) + "];\nr = xcorr(x, y);"
    got = vec(run(tmp_path, body))
    n = max(x.size, y.size)
    xp, yp = np.concatenate([x, np.zeros(n - x.size)]), np.concatenate([y, np.zeros(n - y.size)])
    want = []
    for m in range(-(n - 1), n):
        acc = 0.0
        for k in range(n):
            if 0 <= k + m < n:
                acc += xp[k + m] * yp[k]
        want.append(acc)
    assert got.shape == (2 * n - 1,) and np.max(np.abs(got - np.array(want))) < 1e-12
