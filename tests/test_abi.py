"""The C-ABI shared library loads and exports every symbol include/qasr.h declares; the packed-blob structs
have the sizes pack.py serialises (checked with a gcc-compiled probe).  No compute calls: runs without a GPU."""
import ctypes
import os
import re
import struct
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, 'include', 'qasr.h')


def _declared_symbols():
    src = open(HDR).read()
    return sorted(set(re.findall(r'\b(qasr_[a-z_0-9]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    from qasr import build, engine
    lib = ctypes.CDLL(build.build_native())
    declared = _declared_symbols()
    assert set(declared) == set(engine.SYMBOLS), set(declared) ^ set(engine.SYMBOLS)
    for sym in declared:
        assert hasattr(lib, sym), sym
    lib.qasr_version.restype = ctypes.c_char_p
    assert b'gfx950' in lib.qasr_version()
    lib.qasr_frontend_frames.argtypes = [ctypes.c_int, ctypes.c_int]
    assert lib.qasr_frontend_frames(80000, 16) == 512 and lib.qasr_frontend_frames(80000, 0) == 501


def test_struct_sizes_match_packer(tmp_path):
    probe = tmp_path / 'probe.c'
    probe.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "qasr.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                     'sizeof(qasr_blob_header),sizeof(qasr_tensor_desc),sizeof(qasr_op_desc),sizeof(qasr_out),'
                     'sizeof(qasr_pane),sizeof(qasr_domain_desc),sizeof(qasr_sep_layer_args),'
                     'offsetof(qasr_sep_layer_args, outs),offsetof(qasr_sep_layer_args, racc),'
                     'sizeof(qasr_engine_opts),offsetof(qasr_engine_opts, tile_frames),offsetof(qasr_engine_opts, retired_persistent),'
                     'sizeof(qasr_dyn_view),offsetof(qasr_dyn_view, residue_hi));return 0;}\n')
    exe = tmp_path / 'probe'
    subprocess.run(['gcc', '-I', os.path.join(ROOT, 'include'), str(probe), '-o', str(exe)], check=True)
    sizes = list(map(int, subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()))
    assert sizes[0] == struct.calcsize('<10I5Q')
    assert sizes[1] == struct.calcsize('<IIIIii')
    assert sizes[3] == struct.calcsize('<iiiIQd') and sizes[4] == struct.calcsize('<iIQQQQ')
    assert sizes[2] == struct.calcsize('<IIiIIIIIIIQQQQiifI') + 3 * sizes[3] + 12 * sizes[4]
    assert sizes[5] == struct.calcsize('<iIIII3I')
    from qasr import engine                                   # the ctypes mirror of the operator-level argument block
    assert sizes[6] == ctypes.sizeof(engine.SepLayerArgs)
    assert sizes[7] == engine.SepLayerArgs.outs.offset and sizes[8] == engine.SepLayerArgs.racc.offset
    assert sizes[9] == ctypes.sizeof(engine.EngineOpts) == 64      # the engine's launch-plan options (qasr_engine_create_ex)
    assert sizes[10] == engine.EngineOpts.tile_frames.offset and sizes[11] == engine.EngineOpts.retired_persistent.offset
    from qasr import dynamic                                  # a float tensor as integers x scales (+ division residue)
    assert sizes[12] == ctypes.sizeof(dynamic.DynView) and sizes[13] == dynamic.DynView.residue_hi.offset


def test_engine_opts_defaults_and_validation():
    """qasr_engine_default_opts fills the struct; create_ex validates the options before any HIP call."""
    from qasr import build, engine
    lib = ctypes.CDLL(build.build_native())
    lib.qasr_last_error.restype = ctypes.c_char_p
    o = engine.EngineOpts()
    lib.qasr_engine_default_opts(ctypes.byref(o))
    assert o.struct_size == ctypes.sizeof(engine.EngineOpts) and o.tile_frames == 0 and o.sep_gen == 0 and o.retired_persistent == 0
    assert (o.fuse_dw, o.fuse_stem, o.fuse_decoder, o.res_tile128, o.dense_tile128) == (-1,) * 5 and o.graph == 0
    h = ctypes.c_void_p()
    blob = ctypes.create_string_buffer(256)
    lib.qasr_engine_create_ex.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    o.tile_frames = 48
    assert lib.qasr_engine_create_ex(blob, 256, 0, ctypes.byref(o), ctypes.byref(h)) == 1 and b'tile_frames' in lib.qasr_last_error()
    o.tile_frames, o.struct_size = 128, 4096
    assert lib.qasr_engine_create_ex(blob, 256, 0, ctypes.byref(o), ctypes.byref(h)) == 1 and b'struct_size' in lib.qasr_last_error()
    o.tile_frames, o.struct_size, o.retired_persistent = 128, ctypes.sizeof(engine.EngineOpts), 1   # a retired option is refused loudly
    assert lib.qasr_engine_create_ex(blob, 256, 0, ctypes.byref(o), ctypes.byref(h)) == 4 and b'retired' in lib.qasr_last_error()
    o.retired_persistent = 0
    o.struct_size = 16                                         # an older, shorter struct is accepted (then the blob is checked)
    assert lib.qasr_engine_create_ex(blob, 256, 0, ctypes.byref(o), ctypes.byref(h)) == 2 and b'magic' in lib.qasr_last_error()
    assert lib.qasr_engine_create_ex(blob, 256, 0, None, ctypes.byref(h)) == 2      # NULL options = defaults


def test_engine_create_rejects_garbage_without_touching_gpu():
    from qasr import build
    lib = ctypes.CDLL(build.build_native())
    lib.qasr_last_error.restype = ctypes.c_char_p
    h = ctypes.c_void_p()
    buf = ctypes.create_string_buffer(256)
    rc = lib.qasr_engine_create(buf, ctypes.c_size_t(256), 0, 0, ctypes.byref(h))
    assert rc == 2 and b'magic' in lib.qasr_last_error()      # QASR_ERR_BLOB before any HIP call
    assert lib.qasr_engine_create(None, ctypes.c_size_t(0), 0, 0, ctypes.byref(h)) == 1
