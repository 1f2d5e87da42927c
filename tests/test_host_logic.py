"""CPU tests of the host-side mirror: callbacks, bookkeeping, .h5 I/O, MIDI bytes, C-ABI surface, CLIs."""
import ctypes
import json
import os
import re
import subprocess
import types

import numpy as np
import pytest

import clvae_amd  # noqa: F401
from clvae_amd import _lib
from clvae_amd.keras_like import Variable, get_value
from clvae_amd.utils import h5io, midi_utils
from clvae_amd.utils import model_utils as MU
from helpers import ROOT


# ------------------------------------------------------------------ C ABI --
def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "clvae.h")).read()
    declared = set(re.findall(r"\b(clv_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), "libclvae_hip.so does not export " + name
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    # the library, the header it was built from and the ctypes binding agree on the ABI version (a library of another
    # round resolves the same symbols with other argument lists: the binding refuses it at load time)
    hdr_version = int(re.search(r"#define\s+CLV_ABI_VERSION\s+(\d+)", hdr).group(1))
    assert _lib.lib().clv_version() == hdr_version == _lib.ABI_VERSION


def test_binding_refuses_a_library_of_another_abi_version(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "ABI_VERSION", _lib.ABI_VERSION + 1)
    with pytest.raises(_lib.ClvError, match="ABI version"):
        _lib.lib()
    monkeypatch.undo()
    assert _lib.lib().clv_version() == _lib.ABI_VERSION


def test_no_gpu_fails_loudly():
    if _lib.lib().clv_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.ClvError):
        _lib.require_gpu()
    from clvae_amd.cl_vae.model import get_model
    with pytest.raises(_lib.ClvError):
        get_model(4, 88, (88, 2), (88, 2), 'adam-wn')          # no CPU fallback: building a model needs the device


def test_no_valu_instruction_hides_inside_inline_asm():
    """gfx940/gfx950: a VALU instruction that reads a transcendental instruction's result needs a wait state the compiler
    only inserts around instructions it emits itself; an asm statement holding a v_* instruction once put wrong states
    into the sigmoid-gate pair kernel (csrc/lstm_common.h, Sel4::pick; tools/probes/trans_hazard_asm.hip).  csrc/ keeps
    its asm statements to waits, barriers, clocks, empty register pins and one VMEM load."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    found = []
    for path in sorted(glob.glob(os.path.join(root, "classifying-vae-lstm_amd", "csrc", "*.h*"))):
        src = re.sub(r"//[^\n]*", "", open(path).read())
        for m in re.finditer(r"\basm\s*(?:volatile)?\s*\(\s*((?:\"(?:[^\"\\]|\\.)*\"\s*)+)", src):
            text = "".join(re.findall(r"\"((?:[^\"\\]|\\.)*)\"", m.group(1)))
            for ins in re.split(r"\\n|\\t|;", text):
                if re.match(r"\s*(v_|ds_)\w+", ins):
                    found.append((os.path.basename(path), ins.strip()))
    assert not found, found


def test_no_bit_cast_of_a_vector_element():
    """`__builtin_bit_cast(float, v.y)` on an ELEMENT of an ext_vector_type value reads element 0 with this clang (the
    front end takes the vector's address for the cast: every lane got ki for kf, kg and ko in round 5's first build of the
    wide record loads of csrc/lstm_mx.hip).  Elements are copied to scalars first."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    found = []
    for path in sorted(glob.glob(os.path.join(root, "classifying-vae-lstm_amd", "csrc", "*.h*"))):
        src = re.sub(r"//[^\n]*", "", open(path).read())
        for m in re.finditer(r"__builtin_bit_cast\(\s*[\w ]+,\s*[\w\]\[]+\.[xyzw]\s*\)", src):
            found.append((os.path.basename(path), m.group(0)))
    assert not found, found


def test_every_entry_point_survives_null_and_zero_arguments():
    """Error behaviour of the C ABI (include/clvae.h: "return 0 / negative CLV_E* / positive hipError_t", nothing crashes):
    every exported function called with NULL pointers, zero sizes and zero-filled structs, each in a process of its own --
    a compute entry point answers with an error code before it touches its arguments or the device, a size query with a
    number.  (Round 5 found five size queries that divided by zero here.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prog = (
        "import sys, ctypes as C\n"
        "sys.path.insert(0, %r)\n"
        "import clvae_amd\n"
        "from clvae_amd import _lib\n"
        "L = _lib.lib()\n"
        "for n in sorted(_lib.SIGNATURES):\n"
        "    res, args = _lib.SIGNATURES[n]\n"
        "    vals = []\n"
        "    for a in args:\n"
        "        if a in (C.c_void_p, C.c_char_p): vals.append(None)\n"
        "        elif a in (C.c_float, C.c_double): vals.append(0.0)\n"
        "        elif isinstance(a, type) and issubclass(a, C.Structure): vals.append(a())\n"
        "        elif hasattr(a, '_type_') and isinstance(getattr(a, '_type_'), type): vals.append(None)\n"
        "        else: vals.append(0)\n"
        "    print('CALL', n, flush=True)\n"
        "    r = getattr(L, n)(*vals)\n"
        "    print('RET', n, r if not isinstance(r, bytes) else 0, flush=True)\n" % root)
    r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=300)
    calls = [l.split()[1] for l in r.stdout.splitlines() if l.startswith("CALL")]
    rets = {l.split()[1]: int(l.split()[2]) for l in r.stdout.splitlines() if l.startswith("RET")}
    assert r.returncode == 0, "crashed in %s (exit code %d)" % (calls[-1] if calls else "?", r.returncode)
    assert set(rets) == set(_lib.SIGNATURES)
    compute = [n for n in rets if not n.endswith(("_bytes", "_supported", "_splits", "_floats", "_split", "_version", "_count",
                                                  "_string", "_enable", "_collect", "_scope", "_destroy"))
               and not n.startswith(("clv_splitk_reduce_multi", "clv_graph_", "clv_prof_"))]
    for n in compute:
        assert rets[n] != 0, n             # an error code, not "ok"
    for n in rets:
        if n.endswith(("_bytes", "_splits")):
            assert rets[n] >= 0, n


def test_adam_plan_layout_on_host():
    L = _lib.lib()
    tab = (_lib.ParamDesc * 3)(_lib.ParamDesc(0, 200, 88, 0, 1, 0), _lib.ParamDesc(17600, 1, 88, 0, 0, 0),
                               _lib.ParamDesc(17688, 88, 3, 88, 1, 0))
    nb = L.clv_adam_wn_plan_bytes(tab, 3)
    assert nb > 0 and nb % 16 == 0
    blob = (ctypes.c_uint8 * nb)()
    assert L.clv_adam_wn_plan_build(tab, 3, blob) == 0
    assert L.clv_adam_wn_workspace_bytes(tab, 3) > 0
    assert L.clv_gemm_auto_split(88, 352, 32768) > 1 and L.clv_gemm_auto_split(32768, 352, 88) == 1
    assert L.clv_gemm_f32(0, 0, 0, 4, 4, ctypes.c_float(1), None, 4, None, 4, ctypes.c_float(0), None, 4, None, 0, None,
                          1, None, 0, None, None) == -1       # CLV_EINVAL before any device work
    assert L.clv_lstm_seq_fwd(4, 4, 64, 0, None, None, None, None, None, None, None, None, None, None, None) == -1


# --------------------------------------------------------------- callbacks --
class _FakeModel:
    def __init__(self):
        self.stop_training = False
        self.saved = []

    def save_weights(self, path, overwrite=True):
        self.saved.append(path)


def test_anneal_loss_weight_schedule(capsys):
    v = Variable(0.1)
    cb = MU.AnnealLossWeight(v, name="kl_weight", final_value=1.0, n_epochs=4)
    vals = []
    for e in range(6):
        cb.on_epoch_begin(e)
        vals.append(get_value(v))
    np.testing.assert_allclose(vals, [0.1, 0.325, 0.55, 0.775, 1.0, 1.0])
    assert "+++++ kl_weight: 0.1" in capsys.readouterr().out
    s = MU.AnnealLossWeight(Variable(0.0), n_epochs=10, slope=8)
    assert abs(s.next_weight(0.5) - 0.5) < 1e-12


def test_early_stopping_is_doubled_like_the_reference():
    """get_callbacks appends the same early-stop object twice (utils/model_utils.py:155,157), so with
    patience=5 training stops at the 3rd consecutive non-improving epoch."""
    args = types.SimpleNamespace(model_dir='/tmp', run_name='r', log_dir='/tmp')
    cbs = MU.get_callbacks(args, patience=5, min_epoch=1)
    assert cbs[1] is cbs[2] and isinstance(cbs[0], MU.ModelCheckpointAfterEpoch)
    m = _FakeModel()
    for c in cbs:
        c.set_model(m)
        c.on_train_begin({})
    losses = [9.0, 5.0, 4.0, 4.5, 4.6, 4.7, 4.8]
    stopped_at = None
    for e, l in enumerate(losses):
        for c in cbs:
            c.on_epoch_end(e, {'val_loss': l})
        if m.stop_training:
            stopped_at = e
            break
    assert stopped_at == 5                      # epochs 3,4,5 are the three non-improving ones
    assert m.saved == ['/tmp/r.h5', '/tmp/r.h5']  # epoch 0 is below min_epoch; 1 and 2 improve


def test_to_categorical_and_numpy_helpers():
    np.testing.assert_array_equal(MU.to_categorical([1, 0, 2], 3), np.eye(3)[[1, 0, 2]])
    assert MU.to_categorical(3, 5).shape == (1, 5)          # scalar -> (1, n): how w_val gets its batch axis
    v = np.log(np.array([[1., 2.], [3., 4.]]))
    np.testing.assert_allclose(MU.logsumexp(v), np.log([4., 6.]))
    np.testing.assert_allclose(MU.logmeanexp(v), np.log([2., 3.]))
    opt, was = MU.init_adam_wn('adam-wn')
    assert was and opt.name == 'adam-wn' and opt.lr == 0.001
    assert MU.init_adam_wn('adam') == ('adam', False)


def test_save_model_in_pieces(tmp_path):
    class M:
        def to_yaml(self):
            return "a: 1\n"
    args = types.SimpleNamespace(model_dir=str(tmp_path), run_name='run1', n_classes=np.int64(2), latent_dim=4,
                                 optimizer='adam-wn', use_x_prev=True)
    MU.save_model_in_pieces(M(), args)
    d = json.load(open(tmp_path / 'run1.json'))
    assert d['n_classes'] == 2 and d['use_x_prev'] is True and d['optimizer'] == 'adam-wn'
    assert open(tmp_path / 'run1.yaml').read() == "a: 1\n"


# -------------------------------------------------------------------- .h5 --
def _layers(rng):
    f = lambda *s: rng.standard_normal(s).astype(np.float32)
    return [('current', [], []), ('hW', ['kernel', 'bias'], [f(32, 8), f(8)]), ('W', [], []),
            ('encoder_h', ['kernel', 'recurrent_kernel', 'bias'], [f(18, 32), f(8, 32), f(32)]),
            ('Z_mean', ['kernel', 'bias'], [f(8, 1), f(1)])]


def test_h5_roundtrip(tmp_path):
    layers = _layers(np.random.default_rng(0))
    path = str(tmp_path / "w.h5")
    h5io.save_keras_weights(path, layers)
    back = h5io.load_keras_weights(path)
    assert [n for n, _ in back] == [l[0] for l in layers]
    for (_, _, arrs), (_, got) in zip(layers, back):
        assert len(arrs) == len(got)
        for a, b in zip(arrs, got):
            np.testing.assert_array_equal(a, b)
    with pytest.raises(ValueError):
        open(tmp_path / "bad.h5", 'wb').write(b'not hdf5' * 20)
        h5io.load_keras_weights(str(tmp_path / "bad.h5"))


CONDA_PY = "/opt/conda/bin/python3.9"


@pytest.mark.skipif(not os.path.exists(CONDA_PY), reason="h5py is only available in the build container's conda python")
def test_h5_interoperates_with_h5py(tmp_path):
    layers = _layers(np.random.default_rng(1))
    ours, theirs = str(tmp_path / "ours.h5"), str(tmp_path / "theirs.h5")
    h5io.save_keras_weights(ours, layers)
    code = r'''
import sys, h5py, numpy as np
f = h5py.File(sys.argv[1], "r")
names = [n.decode() for n in f.attrs["layer_names"]]
assert names == ["current", "hW", "W", "encoder_h", "Z_mean"], names
assert f.attrs["keras_version"] == b"2.0.0" and f.attrs["backend"] == b"tensorflow"
tot = 0.0
for n in names:
    g = f[n]
    for w in g.attrs["weight_names"]:
        tot += float(np.abs(g[w.decode()][...]).sum())
print("%.6f" % tot)
# the way Keras 2.0.0 save_weights writes a file
g = h5py.File(sys.argv[2], "w")
g.attrs["layer_names"] = [n.encode("utf8") for n in names]
g.attrs["backend"] = b"tensorflow"; g.attrs["keras_version"] = b"2.0.0"
for n in names:
    gg = g.create_group(n)
    src = f[n]
    wn = list(src.attrs["weight_names"])
    gg.attrs["weight_names"] = wn
    for w in wn:
        v = src[w.decode()][...]
        d = gg.create_dataset(w.decode(), v.shape, dtype=v.dtype); d[...] = v * 2
g.close()
'''
    r = subprocess.run([CONDA_PY, "-c", code, ours, theirs], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    expect = sum(float(np.abs(a).sum()) for _, _, arrs in layers for a in arrs)
    assert abs(float(r.stdout.strip()) - expect) < 1e-2
    back = h5io.load_keras_weights(theirs)
    for (_, _, arrs), (_, got) in zip(layers, back):
        for a, b in zip(arrs, got):
            np.testing.assert_array_equal(2 * a, b)


# ------------------------------------------------------------------- MIDI --
def test_midi_bytes_of_a_tiny_roll(tmp_path):
    roll = np.zeros((3, 88))
    roll[0, [39, 43]] = 1       # C4 (60), E4 (64)
    roll[1, [39]] = 1           # C4 held, E4 released
    roll[2, [46]] = 1           # G4 (67) on, C4 off
    data = midi_utils.MidiWriter().dump_sequence_to_midi(roll, str(tmp_path / "t.mid"))
    hdr = b'MThd' + bytes([0, 0, 0, 6, 0, 1, 0, 2, 0x01, 0xE0])
    meta = b'MTrk' + bytes([0, 0, 0, 8]) + bytes([0x00, 0xFF, 0x58, 0x04, 0x04, 0x02, 0x18, 0x08])
    ev = bytes([0x78, 0x90, 60, 100,     # delta 120: note-on C4 (first event carries the frame's tick)
                0x00, 64, 100,           # running status: note-on E4
                0x78, 0x80, 64, 0,       # frame 1: note-off E4
                0x78, 60, 0,             # frame 2: note-off C4 (running status 0x80)
                0x00, 0x90, 67, 100,     # then note-on G4
                0x78, 0x80, 67, 0])      # flush
    trk = b'MTrk' + bytes([0, 0, 0, len(ev)]) + ev
    assert data == hdr + meta + trk
    assert open(tmp_path / "t.mid", 'rb').read() == data
    assert midi_utils.write_varlen(0) == b'\x00' and midi_utils.write_varlen(240) == b'\x81\x70'
    midi_utils.write_sample(roll, str(tmp_path), "s", isHalfAsSlow=True)
    twice = open(tmp_path / "s.mid", 'rb').read()
    assert twice != data and twice.count(b'\x90') == data.count(b'\x90')     # same notes, frames doubled
    withend = midi_utils.MidiWriter().dump_sequence_to_midi(roll, str(tmp_path / "e.mid"), end_of_track=True)
    assert withend.endswith(b'\x00\xFF\x2F\x00')


# ------------------------------------------------------------------- CLIs --
def _opts(parser):
    out = {}
    for a in parser._actions:
        if a.dest == 'help':
            continue
        out[a.option_strings[0] if a.option_strings else a.dest] = a.default
    return out


def test_cli_argument_surfaces_match_the_reference():
    from clvae_amd.cl_vae import sample as vs, train as vt
    from clvae_amd.cl_vrnn import sample as rs, train as rt
    common = {'run_name': None, '--optimizer': 'adam-wn', '--num_epochs': 200, '--original_dim': 88,
              '--intermediate_dim': 88, '--latent_dim': 2, '--class_weight': 1.0, '--w_log_var_prior': 0.0,
              '--do_log': False, '--predict_next': False, '--use_x_prev': False, '--patience': 5, '--kl_anneal': 0,
              '--w_kl_anneal': 0, '--log_dir': '../data/logs', '--model_dir': '../data/models',
              '--train_file': '../data/input/JSB Chorales_Cs.pickle'}
    assert _opts(vt.build_parser()) == dict(common, **{'--batch_size': 100, '--seq_length': 1,
                                                       '--intermediate_class_dim': 88})   # cl_vae/train.py:78-120
    assert _opts(rt.build_parser()) == dict(common, **{'--batch_size': 200, '--seq_length': 16})  # cl_vrnn/train.py:78-117
    assert _opts(vs.build_parser()) == {'run_name': None, '-n': 1, '--use_z_prior': False, '-t': 32, '--infer_w': False,
                                        '--no_x_prev': False, '--sample_dir': '../data/samples',
                                        '--model_dir': '../data/models', '-i': '',
                                        '--train_file': '../data/input/JSB Chorales_Cs.pickle'}  # cl_vae/sample.py:37-59
    assert _opts(rs.build_parser()) == {'run_name': None, '--infer_w': False, '--discrete_w': False, '-t': 32, '-n': 1,
                                        '-c': None, '--sample_dir': '../data/samples', '-i': '',
                                        '--train_file': '../data/input/JSB Chorales_Cs.pickle'}  # cl_vrnn/sample.py:51-70


def test_initializers_distributions():
    from clvae_amd.engine import vrnn_param_shapes
    from clvae_amd.initializers import init_weights
    cfg = dict(D=88, H=88, L=2, T=16, C=10, use_x_prev=True)
    w = init_weights(vrnn_param_shapes(cfg), cfg, seed=0)
    U = w['encoder_h/recurrent_kernel']
    np.testing.assert_allclose(U @ U.T, np.eye(88), atol=1e-5)                  # orthogonal
    b = w['decoder_h/bias']
    assert b[:88].sum() == 0 and (b[88:176] == 1).all() and b[176:].sum() == 0   # unit_forget_bias
    assert abs(w['X_decoded_mean/kernel'].std() - 0.1) < 0.01                   # RandomNormal(0, 0.1)
    k = w['encoder_h/kernel']
    assert abs(np.abs(k).max() - np.sqrt(6.0 / (98 + 352))) < 1e-3              # glorot_uniform limit
    assert (w['hW/bias'] == 0).all()


def test_ctypes_structs_match_the_c_header(tmp_path):
    """Every ctypes.Structure of _lib.py against the struct of include/clvae.h it mirrors: size and the offset of every field,
    as gcc lays the header out (a hand-written mirror that drifts would pass garbage to the kernels without any error)."""
    import ctypes as C
    import subprocess
    lib_py = _lib
    pairs = {'VaeStepOpts': 'clv_vae_step_opts', 'NoiseDraw': 'clv_noise_draw', 'PairPackSrc': 'clv_pair_pack_src',
             'AdamKnownSums': 'clv_adam_known_sums', 'ParamDesc': 'clv_param_desc', 'GemmProb': 'clv_gemm_prob',
             'ReduceJob': 'clv_reduce_job', 'SkinnyProduct': 'clv_skinny_product', 'LabelBwdRider': 'clv_label_bwd_rider',
             'ProfRecord': 'clv_prof_record', 'BatchCursor': 'clv_batch_cursor',
             'WgradProblem': 'clv_wgrad_problem', 'LabelStage': 'clv_label_stage', 'FrameProj': 'clv_frame_proj'}
    # every Structure of the module is covered
    mirrored = {n for n in dir(lib_py) if isinstance(getattr(lib_py, n), type) and issubclass(getattr(lib_py, n), C.Structure)
                and getattr(lib_py, n) is not C.Structure}
    assert mirrored == set(pairs), mirrored ^ set(pairs)
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "clvae.h"', 'int main(void) {']
    for py, cn in pairs.items():
        lines.append('  printf("%s size %%zu\\n", sizeof(%s));' % (py, cn))
        for fname, _ in getattr(lib_py, py)._fields_:
            lines.append('  printf("%s %s %%zu\\n", offsetof(%s, %s));' % (py, fname, cn, fname))
    lines += ['  return 0;', '}']
    src = tmp_path / 'layout.c'
    src.write_text('\n'.join(lines))
    exe = tmp_path / 'layout'
    subprocess.run(['gcc', '-I' + os.path.join(ROOT, 'include'), '-o', str(exe), str(src)], check=True, stdin=subprocess.DEVNULL)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True, stdin=subprocess.DEVNULL).stdout
    for line in out.splitlines():
        py, what, val = line.split()
        cls = getattr(lib_py, py)
        if what == 'size':
            assert C.sizeof(cls) == int(val), (py, C.sizeof(cls), int(val))
        else:
            assert getattr(cls, what).offset == int(val), (py, what, getattr(cls, what).offset, int(val))


def test_ctypes_signatures_have_the_headers_argument_counts():
    """_lib.SIGNATURES against the prototypes of include/clvae.h: same number of arguments per function, pointer arguments
    bound as pointers and 64-bit / float arguments as such (a binding that drops or reorders an argument would still load)."""
    hdr = open(os.path.join(ROOT, "include", "clvae.h")).read()
    hdr = re.sub(r"/\*.*?\*/", " ", hdr, flags=re.S)
    protos = dict(re.findall(r"\b(clv_[a-z0-9_]+)\s*\(([^()]*)\)\s*;", hdr))
    assert set(protos) >= set(_lib.SIGNATURES)
    for name, (restype, argtypes) in _lib.SIGNATURES.items():
        args = [a.strip() for a in protos[name].split(",")]
        if args == ["void"] or args == [""]:
            args = []
        assert len(args) == len(argtypes), (name, len(args), len(argtypes))
        for a, t in zip(args, argtypes):
            is_ptr = "*" in a
            if is_ptr:
                assert t is ctypes.c_void_p or hasattr(t, "contents") or t is ctypes.c_char_p, (name, a, t)
            elif re.match(r"(const\s+)?float\b", a):
                assert t is ctypes.c_float, (name, a, t)
            elif re.match(r"(const\s+)?(size_t|int64_t|uint64_t|long)\b", a):
                assert ctypes.sizeof(t) == 8, (name, a, t)
            elif re.match(r"(const\s+)?(int|int32_t|uint32_t|unsigned)\b", a):
                assert ctypes.sizeof(t) == 4 and t is not ctypes.c_float, (name, a, t)
