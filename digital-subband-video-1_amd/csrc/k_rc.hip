// k_rc.hip -- device-resident average-bitrate rate control (round 4).
//
// The reference chooses every picture's quantiser from the sizes of the packets before it (quality2quant dsv_encoder.c:70-168,
// the statistics of dsv_enc :816-848): an ABR stream is a chain picture -> packet size -> next quantiser.  Until round 3 that
// chain went through the host (code one frame step, fetch the sizes, pick, upload the next tables): one round trip per picture,
// 640 us per 4K 4:4:4 frame step whose bytes take 35 us.  Everything the chain needs exists on the device after k_hz_scan --
// the payload bits, run counts and DC of the three planes (HzPlaneSum) -- and the rest of the packet (header + side information)
// is known to the host before the picture is coded.  So: one thread per picture
//   mode 0 (before the first frame step of a call): pick the quantiser of the stream's first picture of the call;
//   mode 1 (after a frame step's k_hz_scan): size of the packet just coded -> statistics -> quantiser of the stream's NEXT
//          picture in the call, whose job record (region quantisers, smoothing bounds: dsvg_job_set_quant) is rewritten in place.
// The arithmetic is include/dsvg_rc.h, the same statements the C session layer compiles; the host replays it while it assembles
// the packets and refuses a batch whose quantisers differ from the device's (dsv1_enc.c).
#include "dsvg_kernels.hpp"
#include "dsvg_host.hpp"

__global__ __launch_bounds__(64) void k_rc(JobDev *__restrict__ jobs, const RcJobDev *__restrict__ rcj, dsvg_rc_state *__restrict__ state, int d0, int n, int mode)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const int d = d0 + i;
    const RcJobDev r = rcj[d];
    if (r.slot < 0) return;
    dsvg_rc_state e = state[r.slot];
    if (mode == 0) {
        dsvg_job_set_quant(jobs[d], dsvg_rc_pick(&e, jobs[d].isP, r.forced_intra));
    } else {
        JobDev &jb = jobs[d];
        int dc[3];
        unsigned nb[3];
        for (int p = 0; p < 3; p++) {
            dc[p] = jb.psum[p].dc;
            nb[p] = (unsigned)((jb.psum[p].total_bits + 7) >> 3);
        }
        const unsigned len = dsvg_rc_packet_len((unsigned)r.prefix_len, dc, nb);
        jb.psum[0].rc = jb.quant;                           // what the picture was coded with / the packet size: the host checks both
        jb.psum[1].rc = (int)len;
        jb.psum[2].rc = 0;
        dsvg_rc_after(&e, jb.isP, len);
        if (r.next >= 0) dsvg_job_set_quant(jobs[r.next], dsvg_rc_pick(&e, jobs[r.next].isP, rcj[r.next].forced_intra));
    }
    state[r.slot] = e;
}

void launch_rc(hipStream_t st, JobDev *jobs, const RcJobDev *rcj, dsvg_rc_state *state, int d0, int n, int mode)
{
    if (n < 1) return;
    hipLaunchKernelGGL(k_rc, dim3((n + 63) / 64), dim3(64), 0, st, jobs, rcj, state, d0, n, mode);
}
