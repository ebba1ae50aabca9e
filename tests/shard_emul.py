"""In-process emulation of the chunked frame-sharding protocol (visiondepth3d_amd.sharded.ChunkSharder): G sharders -- one per
emulated rank, each on its own backend -- run the protocol's stages in the order the real ranks would, and the point-to-point
plane hand-off / the two all-gathers are done by hand.  Used by the GPU tests (G contexts on one GPU) and, with the oracle
backend, by the CPU tests; the real collectives are covered by tests/test_sharded_gloo.py."""
import torch


class Emu:
    def __init__(self, sharders):
        self.s = list(sharders)
        self.plane = None          # the plane rank G-1 hands to rank 0 for the next step
        self.first = True

    def step(self, loc_f, loc_d, n_valid=None, blank=None):
        """loc_f[g] / loc_d[g]: rank g's own frames / depth planes of this step (only the valid ones are touched).
        Returns outs[g] = muxed frames of rank g's valid own frames."""
        S = self.s
        prev = None
        for g, sh in enumerate(S):
            if g == 0:
                if not self.first and self.plane is not None:
                    sh.b.plane_import(self.plane, True)
            else:
                sh.b.plane_import(prev, True)
            sh.p1_local(loc_f[g], loc_d[g], n_valid)
            prev = sh.b.plane_export().clone()
        self.plane, self.first = prev, False
        q_all = torch.cat([sh.q_local for sh in S])
        for sh in S:
            sh.r1(q_all.clone(), n_valid)
            sh.p3(n_valid)
        m_all = torch.cat([sh.m_local for sh in S])
        outs = []
        for g, sh in enumerate(S):
            sh.r2(m_all.clone(), n_valid, blank)
            outs.append(sh.pixels(None, n_valid, loc_f[g], blank))
        return outs

    def finish(self):
        """ChunkSharder.finish_clip: every rank ends with the last plane."""
        for sh in self.s[:-1]:
            sh.b.plane_import(self.plane, True)

    def run_clip(self, frames, depths, B, blank_frames=()):
        """Whole clip over the emulated ranks -> {frame index: muxed frame}; the last step may be partial."""
        G = len(self.s)
        per, n = G * B, len(frames)
        res = {}
        for base in range(0, n, per):
            nv = min(per, n - base)
            loc_f = [[frames[base + g * B + j] for j in range(B) if base + g * B + j < n] for g in range(G)]
            loc_d = [[depths[base + g * B + j] for j in range(B) if base + g * B + j < n] for g in range(G)]
            blank = [(base + t) in blank_frames for t in range(nv)] if blank_frames else None
            outs = self.step(loc_f, loc_d, nv, blank)
            for g in range(G):
                for j, o in enumerate(outs[g]):
                    res[base + g * B + j] = o
        self.finish()
        return res
