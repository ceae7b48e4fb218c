"""qasr_blob_check (csrc/qasr_blob_check.cpp; the validation qasr_engine_create_ex runs before any HIP call) under
AddressSanitizer + UBSan on the CPU: every packed model of the repo is accepted, and tests/native/blob_fuzz.cpp's
field-by-field hostile mutations and seeded random corruptions are rejected - or, where a mutation leaves the blob valid,
every array the engine would dereference still lies inside the buffer - without a sanitizer report.  No GPU."""
import json
import os
import subprocess

import numpy as np
import pytest

from qasr import pack, synth, topology

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def fuzz_exe(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp('blobfuzz') / 'blob_fuzz')
    subprocess.run(['g++', '-std=c++17', '-O1', '-g', '-fsanitize=address,undefined', '-fno-sanitize-recover=all',
                    '-I', os.path.join(ROOT, 'include'), os.path.join(ROOT, 'q-asr_amd', 'csrc', 'qasr_blob_check.cpp'),
                    os.path.join(ROOT, 'tests', 'native', 'blob_fuzz.cpp'), '-o', exe], check=True)
    return exe


def _blob(golden_dir, name, cfg):
    d = np.load(os.path.join(golden_dir, name + '.npz'))
    meta = json.loads(str(d['meta']))
    sd = synth.make_state_dict(cfg, meta['seed'])
    return pack.pack_model(cfg, sd, d['act_min'], d['act_max'], meta['wbit'], meta['abit'])[0]


@pytest.mark.parametrize('name,cfg,n_random', [('net_miniq_w8a8', topology.mini_quartznet, 1000),
                                               ('net_minij_w8a8', topology.mini_jasper, 1000),
                                               ('net_miniq_w6a6', topology.mini_quartznet, 300)])
def test_blob_check_under_sanitizers(golden_dir, fuzz_exe, tmp_path, name, cfg, n_random):
    blob = _blob(golden_dir, name, cfg())
    path = tmp_path / 'model.blob'
    path.write_bytes(blob)
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=0:abort_on_error=0', UBSAN_OPTIONS='print_stacktrace=1')
    out = subprocess.run([fuzz_exe, str(path), '7', str(n_random)], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert 'ERROR: AddressSanitizer' not in out.stderr and 'runtime error' not in out.stderr, out.stderr[-3000:]
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec['structural'] > 300 and rec['structural_accepted'] == 0, rec
    assert rec['random'] == n_random and rec['random_ok'] + rec['random_rejected'] == n_random, rec
    assert rec['random_rejected'] > n_random // 2, rec        # most table corruptions are caught; the rest are harmless fields


def test_full_size_blobs_are_accepted_and_engine_create_reports_the_field(golden_dir):
    """The checker accepts what pack.py writes for both model families and both bit widths (so the engine's create path is
    unchanged for valid input), and through the C ABI a corrupted offset comes back as QASR_ERR_BLOB naming the op."""
    import ctypes
    from qasr import build
    lib = ctypes.CDLL(build.build_native())
    lib.qasr_blob_check.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
    why = ctypes.create_string_buffer(256)
    blobs = {n: _blob(golden_dir, n, c()) for n, c in (('net_quartznet_w8a8', topology.quartznet15x5),
                                                       ('net_quartznet_w6a6', topology.quartznet15x5),
                                                       ('net_jasper_w8a8', topology.jasper10x5dr))}
    for n, b in blobs.items():
        assert lib.qasr_blob_check(b, len(b), why, 256) == 0, (n, why.value)
    b = bytearray(blobs['net_quartznet_w8a8'])
    hdr = np.frombuffer(bytes(b[:80]), dtype=np.uint32)
    ops_off = int(np.frombuffer(bytes(b[48:56]), dtype=np.uint64)[0])
    op_size = int(hdr[9])
    w_off_at = ops_off + 5 * op_size + 40                     # qasr_op_desc.w_off of op 5
    b[w_off_at:w_off_at + 8] = (len(b)).to_bytes(8, 'little')
    assert lib.qasr_blob_check(bytes(b), len(b), why, 256) == 2 and b'op 5' in why.value and b'weight' in why.value
    h = ctypes.c_void_p()
    lib.qasr_last_error.restype = ctypes.c_char_p
    assert lib.qasr_engine_create(bytes(b), ctypes.c_size_t(len(b)), 0, 0, ctypes.byref(h)) == 2    # before any HIP call
    assert b'op 5' in lib.qasr_last_error()
