"""CPU-side checks: the C-ABI library builds for gfx950, exports every symbol the headers declare, refuses to run
without a GPU (no CPU fallback), and the host-side index arithmetic of the sharded sweep is self-consistent."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b(gpsx_\w+|gps_\w+|correlation_search|acquisition_\w+|signal_capture_\w+)\s*\(", text))
    return {n for n in names if not n.endswith("_t")}


def test_library_exports_every_declared_symbol(lib_path):
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib_path], text=True)
    exported = {line.split()[-1] for line in out.splitlines() if line.strip()}
    declared = _declared_functions("gpsx.h") | _declared_functions("gpsx_compat.h")
    declared -= {"gpsx_key_energy", "gpsx_key_fine_phase"}        # static inline helpers
    assert len(declared) > 35
    missing = sorted(declared - exported)
    assert not missing, missing
    for var in ("tmp_prn_data", "tmp_data_i", "tmp_data_q"):      # PM/GPS/common_ram.c:3-5
        assert var in exported


def test_library_carries_gfx950_code_object(lib_path):
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--list", "--type=o", f"--input={lib_path}"],
                         capture_output=True, text=True)
    blob = open(lib_path, "rb").read()
    assert b"gfx950" in blob and b"k_acq" in blob
    assert b"gfx942" not in blob and b"sm_" not in blob[:0]      # one target, no fallbacks
    del out


def test_no_cpu_fallback_without_gpu(lib_path):
    """On a box without a GPU the engine must refuse to start (GPSX_ENODEV); on a GPU box this is skipped."""
    lib = ctypes.CDLL(lib_path)
    h = ctypes.c_void_p()
    rc = lib.gpsx_create(ctypes.byref(h), 0, None)
    if rc == 0:
        lib.gpsx_destroy(h)
        pytest.skip("a GPU is present")
    assert rc == -19
    from stm32f4_sdr_gps_amd import capi
    with pytest.raises(capi.GpsxError):
        capi.Engine(0)


def test_product_sources_never_reference_the_oracle():
    pkg = os.path.join(ROOT, "stm32f4_sdr_gps_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".cpp", ".hpp", ".h", "Makefile")):
                text = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "liboracle" not in text and "pyoracle" not in text and "gpsx_oracle" not in text, fn
    for fn in os.listdir(os.path.join(ROOT, "include")):
        assert "oracle" not in open(os.path.join(ROOT, "include", fn)).read().lower().replace("oracle/", "")


def test_compat_struct_layout_matches_reference_header_when_present():
    ref = "/root/reference/Firmware/project_main"
    if not os.path.isdir(ref):
        pytest.skip("reference tree not present")
    probe = r'''
#include <stdio.h>
#include <stddef.h>
#include HDR
int main(void){
 printf("%zu %zu %zu ", sizeof(gps_acq_t), sizeof(gps_tracking_t), sizeof(gps_ch_t));
 printf("%zu %zu %zu %zu %zu ", offsetof(gps_ch_t,tracking_data), offsetof(gps_ch_t,nav_data), offsetof(gps_ch_t,obs_data), offsetof(gps_ch_t,eph_data), offsetof(gps_ch_t,prn_code));
 printf("%zu %zu %zu %zu %zu %zu ", offsetof(gps_acq_t,found_code_phase), offsetof(gps_acq_t,state), offsetof(gps_acq_t,code_phase_histogram), offsetof(gps_acq_t,start_timestamp), offsetof(gps_acq_t,hist_ratio), offsetof(gps_acq_t,given_freq_offset_hz));
 printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\n", offsetof(gps_tracking_t,if_freq_offset_hz), offsetof(gps_tracking_t,if_freq_accum), offsetof(gps_tracking_t,pre_track_phases), offsetof(gps_tracking_t,prev_track_timestamp), offsetof(gps_tracking_t,code_phase_fine), offsetof(gps_tracking_t,fll_old_i), offsetof(gps_tracking_t,pll_check_buf), offsetof(gps_tracking_t,snr_value), offsetof(gps_tracking_t,state));
 printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(gps_nav_data_t), offsetof(gps_nav_data_t,old_swap_time), offsetof(gps_nav_data_t,inv_polarity_flag), offsetof(gps_nav_data_t,word_buf), offsetof(gps_nav_data_t,word_cnt), offsetof(gps_nav_data_t,old_D30), offsetof(gps_nav_data_t,word_detection_timestamp), offsetof(gps_nav_data_t,first_subframe_time), offsetof(gps_nav_data_t,new_subframe_flag), offsetof(gps_nav_data_t,subframe_data));
 printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(eph_t), sizeof(sdreph_t), sizeof(gps_obs_data_t), offsetof(eph_t,week), offsetof(eph_t,toc), offsetof(eph_t,M0), offsetof(eph_t,cus), offsetof(eph_t,fit), offsetof(eph_t,f2), offsetof(sdreph_t,week_gpst), offsetof(sdreph_t,sub_cnt), offsetof(sdreph_t,received_mask_proc));
 return 0; }
'''
    import tempfile
    outs = []
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "p.c")
        open(src, "w").write(probe)
        for hdr, inc in (('"gps_misc.h"', [f"-I{ref}", f"-I{ref}/GPS", f"-I{ref}/GPS/RTK"]), ('"gpsx_compat.h"', [f"-I{ROOT}/include"])):
            exe = os.path.join(td, "p")
            subprocess.check_call(["gcc", "-w", f"-DHDR={hdr}", *inc, src, "-o", exe])
            outs.append(subprocess.check_output([exe], text=True))
    assert outs[0] == outs[1], outs


def test_sharding_partition_and_key_packing():
    from stm32f4_sdr_gps_amd import sharding
    for (ns, npn, nd) in [(16, 32, 21), (2, 20, 5), (1, 9, 29)]:
        for world in (1, 2, 3, 8):
            total = np.zeros((ns, npn, nd), np.int32)
            for r in range(world):
                total += sharding.owned_mask(ns, npn, nd, r, world)
            assert (total == 1).all()
    # ties between bit shifts and offsets resolve to the lowest fine phase
    mv = np.array([[7, 7, 9, 9, 1, 0, 0, 0]])
    ph = np.array([[3, 2, 5, 4, 0, 0, 0, 0]])
    k = sharding.pack_keys(mv, ph)
    e, f = sharding.unpack_keys(k)
    assert int(e[0]) == 9 and int(f[0]) == 8 * 4 + 3
    # weak-scaling balance of the bench default (16 searches per GPU): every rank gets the same number of units
    for world in (1, 2, 4, 8):
        counts = [int(sharding.owned_mask(16 * world, 32, 21, r, world)[:, ::8, :].sum()) for r in range(world)]
        assert len(set(counts)) == 1 and counts[0] == 16 * 4 * 21
    # BASELINE.json configs[3] as written: ONE 32 PRN x 21 Doppler search on 8 GPUs = 84 units, 11 or 10 per rank
    # (SURVEY.md 8(e)); a unit is one 8-PRN group x one Doppler bin
    assert sharding.GROUP == 8 and int(sharding.unit_table(1, 32, 21).max()) == 83
    counts = [int(sharding.owned_mask(1, 32, 21, r, 8)[:, ::8, :].sum()) for r in range(8)]
    assert sorted(counts) == [10, 10, 10, 10, 11, 11, 11, 11]
    assert max(counts) / (84 / 8) < 1.05
    # ... and at most two (search, Doppler) pairs per rank are split between ranks (whole 32-PRN clusters elsewhere)
    for r in range(8):
        m = sharding.owned_mask(1, 32, 21, r, 8)[0, ::8, :]          # [group, dopp]
        partial = int(((m.sum(axis=0) > 0) & (m.sum(axis=0) < 4)).sum())
        assert partial <= 2
    # the ownership rule written out by hand for a small grid: unit = (search * n_dopp + dopp) * n_groups + prn // 8,
    # rank 1 of 3 owns units [U / 3, 2 U / 3) with U = 2 * 3 * 2
    m = sharding.owned_mask(2, 10, 3, 1, 3)
    for s in range(2):
        for p in range(10):
            for d in range(3):
                assert m[s, p, d] == (4 <= (s * 3 + d) * 2 + p // 8 < 8)


def test_public_headers_are_plain_c(tmp_path):
    """include/*.h are the drop-in boundary of a C code base: strict C99 and C++11 must both take them, warnings as errors."""
    src = tmp_path / "hdr_check.c"
    src.write_text('#include "gpsx.h"\n#include "gpsx_compat.h"\n'
                   "int main(void) { gpsx_acq_grid_t g; gps_ch_t c; (void)g; (void)c; return (int)sizeof(gpsx_peak_t) - 16; }\n")
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", inc, "-fsyntax-only", str(src)])
    subprocess.check_call(["g++", "-std=c++11", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", inc, "-fsyntax-only",
                           "-x", "c++", str(src)])


def test_reference_step_sources_relink_against_libgpsx_unmodified(lib_path, tmp_path):
    """INTEGRATION.md tier 1, "relink, no source change": the reference's own acquisition.c and tracking.c, compiled
    where they lie, link against libgpsx.so, and every correlator primitive they call (PM/GPS/gps_misc.h:195-216) and
    the three scratch buffers (PM/GPS/common_ram.h:12-14) resolve to the library -- not to a stray reference object.
    Container-only (needs /root/reference); nothing is copied and nothing is run (no GPU here)."""
    ref = "/root/reference/Firmware/project_main"
    if not os.path.isdir(ref):
        pytest.skip("reference tree not present")
    inc = [f"-I{ref}", f"-I{ref}/GPS", f"-I{ref}/GPS/RTK"]
    objs = []
    for name in ("acquisition", "tracking"):
        obj = tmp_path / f"{name}.o"
        subprocess.check_call(["gcc", "-w", "-O2", "-fno-strict-aliasing", "-c", *inc, f"{ref}/GPS/{name}.c", "-o", str(obj)])
        objs.append(str(obj))
    main_c = tmp_path / "main.c"
    main_c.write_text(
        "#include <stdint.h>\n#include \"gps_misc.h\"\n#include \"acquisition.h\"\n#include \"tracking.h\"\n"
        "gps_ch_t ch[GPS_SAT_CNT]; uint8_t blk[2048];\n"
        "int main(void){ gps_fill_summ_table(); gps_channell_prepare(&ch[0]); acquisition_start_channel(&ch[0]);\n"
        " acquisition_process(ch, blk); gps_tracking_process(&ch[0], blk, 0); return 0; }\n")
    exe = tmp_path / "relinked"
    libdir = os.path.dirname(lib_path)
    subprocess.check_call(["gcc", "-w", *inc, str(main_c), *objs, "-o", str(exe), f"-L{libdir}", "-lgpsx", "-lm",
                           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    undefined = {l.split()[-1] for l in subprocess.check_output(["nm", "-u", str(exe)], text=True).splitlines() if l.strip()}
    lib_defined = {l.split()[-1] for l in subprocess.check_output(["nm", "-D", "--defined-only", lib_path], text=True).splitlines()
                   if l.strip()}
    primitives = {"gps_fill_summ_table", "gps_channell_prepare", "gps_correlation8", "gps_correlation_iq",
                  "correlation_search", "gps_shift_to_zero_freq", "gps_shift_to_zero_freq_track",
                  "gps_generate_prn_data2", "gps_rewind_if_phase"}
    buffers = {"tmp_prn_data", "tmp_data_i", "tmp_data_q"}
    strip = lambda s: s.split("@")[0]
    undefined = {strip(s) for s in undefined}
    assert primitives <= undefined, sorted(primitives - undefined)          # imported, not defined by the reference objects
    # data objects used from an executable are bound by copy relocation: undefined in the reference's objects, a
    # R_X86_64_COPY against the library's definition in the linked program
    obj_undef = {strip(l.split()[-1]) for o in objs for l in subprocess.check_output(["nm", "-u", o], text=True).splitlines()
                 if l.strip()}
    assert buffers <= obj_undef, sorted(buffers - obj_undef)
    relocs = subprocess.check_output(["readelf", "-r", "-W", str(exe)], text=True)
    for b in buffers:
        assert re.search(r"R_X86_64_(COPY|GLOB_DAT)\s+\S+\s+" + b + r"\b", relocs), b
    assert primitives | buffers <= lib_defined
    # the step logic in this executable IS the reference's (its objects win over the library's same-named exports)
    own = {l.split()[-1] for l in subprocess.check_output(["nm", "--defined-only", str(exe)], text=True).splitlines() if l.strip()}
    assert {"acquisition_process", "gps_tracking_process"} <= own
    needed = subprocess.check_output(["readelf", "-d", str(exe)], text=True)
    assert "libgpsx.so" in needed


def test_loop_state_conversion_round_trips_and_touches_only_the_loops_fields(lib_path):
    """gpsx_loop_state_from_channel / gpsx_loop_state_to_channel (host code): a tracking channel's record -> the 120-byte
    device-resident loop state -> back.  Everything the loops own survives the round trip bit for bit; nothing else of the
    record (acquisition result, word layer, observations, ephemeris, PRN code) is written."""
    import ctypes as C

    import numpy as np

    from stm32f4_sdr_gps_amd import capi
    lib = C.CDLL(lib_path)
    lib.gpsx_loop_state_from_channel.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    lib.gpsx_loop_state_to_channel.argtypes = [C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(3)
    rec = rng.integers(0, 256, 1688, dtype=np.uint8)
    rec[60 + 80:60 + 84] = np.frombuffer(np.float32(1234.5).tobytes(), np.uint8)     # (keep the floats finite)
    for off in (4, 92, 96, 104, 132, 144):
        rec[60 + off:60 + off + 4] = np.frombuffer(np.float32(rng.uniform(-100, 100)).tobytes(), np.uint8)
    rec[664] = 17
    st = np.zeros(1, capi.LOOP_DTYPE)
    lib.gpsx_loop_state_from_channel(rec.ctypes.data, 99, st.ctypes.data)
    assert int(st["prn"][0]) == 17 and int(st["rng"][0]) == 99 and float(st["code_phase_fine"][0]) == 1234.5
    assert int(st["found_freq_offset_hz"][0]) == int(rec[2:4].view("<i2")[0])
    assert int(st["if_freq_accum"][0]) == int(rec[60 + 8:60 + 12].view("<u4")[0])
    assert int(st["old_swap_time"][0]) == int(rec[212 + 4:212 + 8].view("<u4")[0]) and int(st["inv_polarity_flag"][0]) == rec[212 + 13]
    back = rec.copy()
    back[60:212] ^= 0xFF                        # scramble what the loops own, then restore it from the state
    back[212:226] ^= 0xFF
    lib.gpsx_loop_state_to_channel(st.ctypes.data, back.ctypes.data)
    st2 = np.zeros(1, capi.LOOP_DTYPE)
    lib.gpsx_loop_state_from_channel(back.ctypes.data, 99, st2.ctypes.data)
    for f in st.dtype.names:
        if f not in ("word_buf", "word_detection_timestamp", "word_cnt", "word_bit_cnt", "inv_preabmle_cnt", "word_flags"):
            assert st[f].tobytes() == st2[f].tobytes(), f    # (those six: the host word layer's own fields, read, never written back)
    n = rec[212:324]
    assert int(st["word_buf"][0]) == sum(int(n[16 + i] & 1) << i for i in range(30))
    assert int(st["word_cnt"][0]) == n[46] and int(st["word_bit_cnt"][0]) == n[47] and int(st["inv_preabmle_cnt"][0]) == n[15]
    assert int(st["word_flags"][0]) == (n[48] & 1) | ((n[49] & 1) << 1) | (4 if n[14] else 0)
    assert int(st["word_detection_timestamp"][0]) == int(n[52:56].copy().view("<u4")[0])
    untouched = np.ones(1688, bool)
    untouched[60:212] = False                   # tracking_data
    untouched[212:226] = False                  # the bit synchroniser's part of nav_data and the polarity flag
    assert np.array_equal(back[untouched], rec[untouched])
    # inside tracking_data the fields the loops do NOT own stay as they were (here: scrambled): code_search_*, pre_track_*,
    # old_code_phase_fine, code_phase_swap_flag, filt_start_time_ms, state (prev_track_timestamp IS the loops': the tick the
    # device last served the channel on -- a host that takes the channel back must see the right elapsed time)
    assert int(st["prev_track_timestamp"][0]) == int(rec[60 + 76:60 + 80].view("<u4")[0])
    for lo, hi in ((0, 4), (12, 76), (84, 89), (136, 140), (148, 152)):
        assert np.array_equal(back[60 + lo:60 + hi], rec[60 + lo:60 + hi] ^ 0xFF), (lo, hi)


def test_product_and_lab_builds_and_no_register_spills(lib_path):
    """lib/libgpsx.so is the product: built without GPSX_LAB, it reads none of the $GPSX_ACQ_* / $GPSX_TRACK_WAVE_FROM knobs that
    force a kernel form (the getenv block is not compiled into it) and cannot carry the wrong-result timing ablations
    (k_acq_mx.hip / gpsx_api.hip refuse those macros without GPSX_LAB).  lib/libgpsx_lab.so is the same sources with the knobs.
    And no instance of the matrix-core grid kernel spills registers (k_acq_mx<3> sits at 255 VGPRs)."""
    import ctypes as C
    import os
    import subprocess

    from stm32f4_sdr_gps_amd import build
    lab_path = os.path.join(os.path.dirname(lib_path), "libgpsx_lab.so")
    assert C.CDLL(lib_path).gpsx_is_lab_build() == 0 and C.CDLL(lab_path).gpsx_is_lab_build() == 1
    strings = subprocess.check_output(["strings", "-a", lib_path], text=True)
    lab_strings = subprocess.check_output(["strings", "-a", lab_path], text=True)
    for knob in ("GPSX_ACQ_ALGO", "GPSX_ACQ_SEG", "GPSX_ACQ_SPLIT", "GPSX_ACQ_NO_SPLIT", "GPSX_ACQ_MS_MODE", "GPSX_TRACK_WAVE_FROM",
                 "GPSX_MX_EXPERIMENT"):
        assert knob not in strings, knob
    assert "GPSX_ACQ_ALGO" in lab_strings and "GPSX_MX_EXPERIMENT" not in lab_strings
    res = build.check_no_scratch()
    mx = {k: v for k, v in res.items() if "k_acq_mx" in k}
    loops = {k: v for k, v in res.items() if "k_track_loop" in k}
    # six k_acq_mx<MODE> instances and k_acq_mxw (the weighted extension's matrix-core kernel)
    assert len(mx) == 7 and all(v["scratch_bytes"] == 0 and v["vgprs"] <= 256 for v in mx.values())
    assert sum("k_acq_mxw" in k for k in mx) == 1
    # the device tracking loops: no spills, and the two xorshift instantiations (the ones that run at scale) at three waves per SIMD
    assert len(loops) == 4 and all(v["scratch_bytes"] == 0 for v in loops.values())
    assert sorted(v["vgprs"] for v in loops.values())[:2] <= [168, 168]
    # the ablation macros do not compile into a product object
    src = os.path.join(os.path.dirname(os.path.dirname(lib_path)), "csrc", "k_acq_mx.hip")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-fsyntax-only", "-DGPSX_MX_NO_PIECES", src],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "GPSX_LAB" in r.stderr


def test_abi_handshake_accepts_this_header_and_refuses_another_layout(lib_path):
    """gpsx_abi_check (include/gpsx.h): the header's version and record sizes a host was compiled with against the library's.
    The sizes come from a C probe compiled against include/gpsx.h here; 96 bytes was gpsx_loop_state_t up to version 100."""
    import re
    import tempfile
    lib = ctypes.CDLL(lib_path)
    lib.gpsx_abi_check.argtypes = [ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t]
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, "probe.c")
        with open(src, "w") as f:
            f.write('#include <stdio.h>\n#include "gpsx.h"\nint main(void){printf("%d %zu %zu %zu\\n", GPSX_VERSION, '
                    'sizeof(gpsx_loop_state_t), sizeof(gpsx_acq_grid_t), sizeof(gpsx_peak_t));return 0;}\n')
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", os.path.join(tmp, "probe")])
        ver, s_loop, s_grid, s_peak = (int(x) for x in subprocess.check_output([os.path.join(tmp, "probe")], text=True).split())
    assert ver == lib.gpsx_version() == 110 and s_loop == 120 and s_peak == 16
    assert int(re.search(r"#define GPSX_VERSION\s+(\d+)", open(os.path.join(ROOT, "include", "gpsx.h")).read()).group(1)) == ver
    assert lib.gpsx_abi_check(ver, s_loop, s_grid, s_peak) == 0
    assert lib.gpsx_abi_check(100, 96, s_grid, s_peak) == -22          # a host built against the round-4 header
    assert lib.gpsx_abi_check(ver, 96, s_grid, s_peak) == -22
    assert lib.gpsx_abi_check(ver, s_loop, s_grid + 8, s_peak) == -22
