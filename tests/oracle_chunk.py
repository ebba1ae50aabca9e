"""The chunked frame-sharding protocol (visiondepth3d_amd.sharded.ChunkSharder) with the CPU ORACLE as backend: the same
stages as HipChunkBackend (P1 / plane hand-off / R1 / P3 / R2 / pixel pass), built from the exported pieces of
oracle/vd3d_oracle.c in exactly the order vo_render_frame composes them, so that a sharded run can be compared BIT FOR BIT with
the sequential oracle render -- numerics, not a recording fake.  TEST INFRASTRUCTURE (imports oracle/).

Record formats are the backend's own business (the orchestrator only moves fixed-size tensors): q = (q_lo, q_hi) float32,
m = 4 x int64 = {bits(dyn_scale f64), bits(motion f64), bits(s_norm f32) | bits(s1 f32) << 32, 0}.
"""
import ctypes as C
import struct

import numpy as np
import torch

from oracle import oracle as O
from visiondepth3d_amd._abi import ShiftParams, State

F32 = np.float32


def _bits64(x):
    return struct.unpack("<q", struct.pack("<d", float(x)))[0]


def _from64(i):
    return struct.unpack("<d", struct.pack("<q", int(i)))[0]


def _bits32(x):
    return struct.unpack("<I", struct.pack("<f", float(x)))[0]


def _from32(i):
    return struct.unpack("<f", struct.pack("<I", int(i) & 0xFFFFFFFF))[0]


def _normalise(plane, row):
    lo, den, collapse = row[0], row[1], row[2]
    d = np.clip(plane, F32(0), F32(1)).astype(F32)
    if collapse:
        return d
    return np.clip(((d - F32(lo)) / F32(den)).astype(F32), F32(0), F32(1)).astype(F32)


class OracleChunkBackend:
    def __init__(self, params):
        self.p = params
        self.device = torch.device("cpu")
        self.state = State()
        self.ne = params.eye_h * params.eye_w
        self.tdf = np.zeros(self.ne, F32)
        self.slots = {}
        self.norm_row = (F32(0), F32(1), 1, 0)    # (lo, den, collapse, have_prev) in force after the last rendered frame
        self.etab = []
        self.batch_calls = []                     # (stage, frames, step_idx0, slot0) of every batched call (tests look at it)
        L = O.lib()
        L.vo_zero_parallax_raw.argtypes = [C.POINTER(ShiftParams), C.c_int, C.c_float, C.POINTER(C.c_float)]
        L.vo_finish_blank.argtypes = [O._u8p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, O._u8p]
        L.vo_shift_smooth.argtypes = [C.POINTER(State)] + [C.POINTER(C.c_double)] * 3
        self.L = L

    # ---- lifecycle / plane state
    def begin(self, n_slots):
        pass

    def new_clip(self):
        s = self.state
        s.smooth_valid = s.tdf_valid = s.prev_depth_valid = s.focal_valid = 0

    def plane_shape(self):
        return (self.p.eye_h, self.p.eye_w)

    def plane_export(self, out=None):
        return torch.from_numpy(self.tdf.reshape(self.plane_shape()).copy())

    def plane_import(self, plane, valid=True):
        self.tdf[:] = plane.numpy().reshape(-1)
        self.state.tdf_valid = 1 if valid else 0

    # ---- P1: ingest + plane EMA + q.02 / q.98 (vo_render_frame_impl up to vo_temporal_filter)
    def p1(self, frame, depth, step_idx, slot, q_out):
        p = self.p
        fb = frame.numpy()
        ft = O.frame_to_tensor(fb)
        d = depth.numpy()
        dt = (d.astype(F32) / F32(255.0)).astype(F32) if d.dtype == np.uint8 else d.astype(F32)
        cy, cx, ch, cw = p.crop_y, p.crop_x, p.crop_h, p.crop_w
        fe = O.interp_bilinear(ft[:, cy:cy + ch, cx:cx + cw], p.eye_h, p.eye_w, aten_threads=p.aten_sum_threads)
        de = O.interp_bilinear(dt[None, cy:cy + ch, cx:cx + cw], p.eye_h, p.eye_w, aten_threads=p.aten_sum_threads)[0]
        prev_plane = self.tdf.copy()
        de_flat = np.ascontiguousarray(de.reshape(-1))
        self.L.vo_temporal_filter(C.byref(self.state), self.tdf.ctypes.data_as(O._f32p), de_flat.ctypes.data_as(O._f32p), self.ne)
        filt = self.tdf.copy()
        dcl = np.clip(filt, F32(0), F32(1)).astype(F32)
        q_out[0] = O.quantile(dcl, F32(0.02))
        q_out[1] = O.quantile(dcl, F32(0.98))
        self.slots[slot] = dict(fe=fe, filt=filt, prev_plane=prev_plane, frame=fb)

    # The batched entry points of the HIP backend (vd3d_shard2_p1_batch / _p3_batch: consecutive own frames in launches of <= 16): here the
    # same per-frame stages in a loop, cut at the same batch boundary, so that the gloo tests drive the branches of ChunkSharder.p1_local / p3
    # the GPU runs (frame j of the call <-> step index step_idx0 + j, slot slot0 + j, record row j).
    MAX_BATCH = 16

    def p1_batch(self, frames, depths, step_idx0, slot0, q_out):
        assert len(frames) == len(depths) and len(frames) >= 1
        self.batch_calls.append(("p1", len(frames), step_idx0, slot0))
        for j0 in range(0, len(frames), self.MAX_BATCH):
            for j in range(j0, min(len(frames), j0 + self.MAX_BATCH)):
                self.p1(frames[j], depths[j], step_idx0 + j, slot0 + j, q_out[j])

    def p3_batch(self, slot0, step_idx0, n, m_out):
        assert n >= 1
        self.batch_calls.append(("p3", n, step_idx0, slot0))
        for j in range(n):
            self.p3(slot0 + j, step_idx0 + j, m_out[j])

    # ---- R1: DepthPercentileEMA over all frames of the step (vo_percentile_ema_normalize's scalar half)
    def r1(self, q_all):
        st = self.state
        q = q_all.numpy()
        self.etab = [(self.norm_row[0], self.norm_row[1], self.norm_row[2], int(st.prev_depth_valid))]
        for lo, hi in q:
            lo, hi = F32(lo), F32(hi)
            if F32(hi - lo) < F32(1e-5):
                collapse = 1
            else:
                collapse = 0
                if not st.ema_valid:
                    st.ema_lo, st.ema_hi, st.ema_valid = float(lo), float(hi), 1
                else:
                    a, b = F32(0.92), F32(1 - 0.92)
                    st.ema_lo = float(F32(a * F32(st.ema_lo)) + F32(b * lo))
                    st.ema_hi = float(F32(a * F32(st.ema_hi)) + F32(b * hi))
            den = F32(F32(st.ema_hi) - F32(st.ema_lo)) + F32(1e-6)
            self.etab.append((F32(st.ema_lo), den, collapse, 1))

    # ---- P3: everything else that is measured on the frame's own planes
    def p3(self, slot, step_idx, m_out):
        p, sl = self.p, self.slots[slot]
        row, prow = self.etab[step_idx + 1], self.etab[step_idx]
        dn = _normalise(sl["filt"], row)
        sl["dn"] = dn
        scale = O.dynamic_parallax_scale(dn.reshape(p.eye_h, p.eye_w), 0.90, 1.15, aten_threads=p.aten_sum_threads)
        motion = 0.0
        if prow[3]:
            motion = O.motion_metric(_normalise(sl["prev_plane"], prow), dn, aten_threads=p.aten_sum_threads)
        s_norm = O.subject_depth(dn.reshape(p.eye_h, p.eye_w))
        sp = self._literal_shift_params(0.0, 0.0, 0.0)
        s1 = O.pixel_shift(sl["fe"], dn.reshape(1, p.eye_h, p.eye_w), p.warp_w, p.warp_h, sp, State())["dbg"]["s1"]
        m_out[0] = _bits64(scale)
        m_out[1] = _bits64(motion)
        m_out[2] = _bits32(s_norm) | (_bits32(s1) << 32)
        m_out[3] = 0

    def _literal_shift_params(self, fg, mg, bg):
        """render_sbs_3d forwards literals for the pop / lock controls and never forwards parallax_balance (:1284-1331)."""
        sp = ShiftParams()
        C.memmove(C.byref(sp), C.byref(self.p.shift), C.sizeof(ShiftParams))
        sp.fg_shift, sp.mg_shift, sp.bg_shift = fg, mg, bg
        sp.parallax_balance, sp.depth_pop_gamma, sp.depth_pop_mid = 0.8, 0.85, 0.50
        sp.depth_stretch_lo, sp.depth_stretch_hi = 0.05, 0.95
        sp.fg_pop_multiplier, sp.bg_push_multiplier, sp.subject_lock_strength = 1.20, 1.10, 1.00
        sp.aten_threads = self.p.aten_sum_threads   # the N-thread ATen mode covers the scalar tails and the small-output bilinear kernel too
        return sp

    # ---- R2: every remaining recurrence, in frame order (the scalar half of vo_render_frame_impl)
    def r2(self, m_all, own_slots, blank):
        p, st, L = self.p, self.state, self.L
        m = m_all.numpy()
        for t in range(len(own_slots)):
            is_blank = bool(blank[t]) if blank is not None else False
            scale, motion = _from64(m[t, 0]), _from64(m[t, 1])
            s_norm, s1 = _from32(int(m[t, 2]) & 0xFFFFFFFF), _from32((int(m[t, 2]) >> 32) & 0xFFFFFFFF)
            fg, mg, bg = C.c_double(p.shift.fg_shift), C.c_double(p.shift.mg_shift), C.c_double(p.shift.bg_shift)
            L.vo_shift_smooth(C.byref(st), C.byref(fg), C.byref(mg), C.byref(bg))
            fg, mg, bg = fg.value * scale, mg.value * scale, bg.value * scale
            if p.ipd_factor != 0.0 and not is_blank:
                fg, mg, bg = fg * p.ipd_factor, mg * p.ipd_factor, bg * p.ipd_factor
            fw_before = (st.fw_prev_offset, st.fw_frame_counter)
            if not is_blank:   # the FloatingWindowTracker update pixel_shift_cuda performs (:651)
                sp = self._literal_shift_params(fg, mg, bg)
                z = C.c_float()
                if L.vo_zero_parallax_raw(C.byref(sp), p.warp_w, C.c_float(s1), C.byref(z)) and sp.enable_floating_window:
                    L.vo_fw_smooth_offset(C.byref(st), float(z.value), 0.0015)
                focal = L.vo_focal_update(C.byref(st), motion if self.etab[t][3] else 0.0, float(F32(s_norm)))
            else:
                focal = st.focal
            s = F32(s_norm)
            rz = F32(F32(F32(F32(-s) * F32(fg)) + F32(F32(-s) * F32(mg))) + F32(s * F32(bg))) / F32(p.warp_w / 2 + 1e-6)
            sz = L.vo_conv_update(C.byref(st), float(rz))
            bw = side = 0
            if p.shift.enable_floating_window and p.shift.use_subject_tracking:
                raw = int(abs(sz) * p.warp_w * 0.75)
                bw = max(0, min(80, L.vo_bar_ease(C.byref(st), raw)))
                side = 1 if sz > 0.005 else (2 if sz < -0.005 else 0)
            st.prev_depth_valid = 1
            sl = own_slots[t]
            if sl >= 0:
                self.slots[sl].update(fg=fg, mg=mg, bg=bg, fw_before=fw_before, focal=focal, bw=bw, side=side, blank=is_blank)
        last = self.etab[len(own_slots)]
        self.norm_row = (last[0], last[1], last[2], 1)

    # ---- pixel pass of an own frame
    def pixels(self, slot, out=None, blank_frame=None):
        p, sl = self.p, self.slots[slot]
        res = np.empty((p.out_h, p.out_w, 3), np.uint8)
        if blank_frame is not None:
            fb, pf = O._u(blank_frame.numpy())
            assert self.L.vo_finish_blank(pf, p.src_h, p.src_w, C.byref(p), sl["bw"], sl["side"], res.ctypes.data_as(O._u8p)) == 0
            return torch.from_numpy(res)
        st = State()   # pixel_shift_cuda sees the tracker as it was BEFORE this frame and repeats the replayed update
        st.fw_prev_offset, st.fw_frame_counter = sl["fw_before"]
        sp = self._literal_shift_params(sl["fg"], sl["mg"], sl["bg"])
        r = O.pixel_shift(sl["fe"], sl["dn"].reshape(1, p.eye_h, p.eye_w), p.warp_w, p.warp_h, sp, st)
        return torch.from_numpy(O.finish_frame(r["left"], r["right"], sl["dn"].reshape(p.eye_h, p.eye_w), p, sl["focal"], sl["bw"], sl["side"]))
