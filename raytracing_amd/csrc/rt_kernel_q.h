// rt_kernel_q.h -- (RT_QUEUES, experimental) the float render kernel with rays regrouped ACROSS the waves of a workgroup.
// Included at the end of rt_kernels.h, inside namespace rtamd.
//
// render_kernel keeps a ray on its lane from the camera to the end of its path: a lane whose traversal has finished waits
// for the wave's next shading trip (18 of 64 lanes per trip on final_scene), and a shading trip runs with the 47 lanes
// that wait.  Here a lane only TRAVERSES: when its ray reaches the end of the stream, the lane writes the ray and its
// closest hit into a ring buffer in LDS (the hit ring) and takes its next ray from a second ring (the ray ring), whoever
// put it there.  A shading trip takes 64 entries of the hit ring, whoever put them there, runs material::scatter / get_ray
// on all 64 lanes and puts the rays that come out into the ray ring.  Paths do not belong to waves: an entry names the
// accumulator slot (wave, slot, pixel) its sample goes to, the adds are LDS atomics on 64-bit fixed-point cells (order
// does not matter), and a counter per slot says how many of the item's paths are still somewhere in the workgroup.
//
// Rings: RT_QCAP entries of LdsLayout::q_fields dwords, field-major (a wave's accesses are conflict-free); four control
// words each -- tail reserved / tail published / head reserved / head released.  A producer reserves entries with a
// compare-and-swap on `tail reserved`, fills them and publishes IN ORDER (it waits until the publications before its own
// are in: they are a few LDS stores away); a consumer does the same on the head pair.  Nothing waits for ENTRIES, and a
// wave that finds the hit ring full keeps its finished rays on their lanes.  Only a shading trip waits, for SPACE in the
// ray ring, and that wait ends: at most RT_Q_SHADERS waves shade at a time, the others take rays out whenever they have
// idle lanes, and the number of paths in the workgroup is bounded (RT_Q_POP_MAX) below what both rings, full, plus every
// other wave's lanes, all holding finished rays, would be (static_assert in the kernel).
//
// MEASURED SLOWER than render_kernel (116 vs 67 ms for 100 strata of final_scene, frames identical
// bit for bit) and therefore not in the shipped library: what a waiting lane cost is now what an exchange trip costs.
#ifndef RT_KERNEL_Q_H
#define RT_KERNEL_Q_H

#ifndef RT_Q_XMIN // lanes that must wait (finished rays to hand over / idle lanes to fill) before a wave spends an exchange trip
#define RT_Q_XMIN 12
#endif
#ifndef RT_Q_POP_MAX // paths per workgroup: the lanes + what the rings should hold on average
#define RT_Q_POP_MAX (LdsLayout<float>::block + 192)
#endif
#ifndef RT_Q_SHADERS // waves that may shade at the same time
#define RT_Q_SHADERS 4
#endif
#ifndef RT_Q_SHADE_MIN // entries of the hit ring that make a shading trip
#define RT_Q_SHADE_MIN 64
#endif

enum : int { QC_T_RES = 0, QC_T_DONE = 1, QC_H_RES = 2, QC_H_DONE = 3, QC_HIT = 0, QC_RAY = 4, QC_ROOM = 8 /* waves in a shading trip */, QC_POP = 9, QC_PANIC = 10 };
enum : int { A_XCHG = 2, A_NONE = 3 };

RT_DEV unsigned q_ld(unsigned *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
RT_DEV void q_st(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
RT_DEV unsigned q_add(unsigned *p, unsigned v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP); }
RT_DEV bool q_cas(unsigned *p, unsigned &expect, unsigned want) {
    return __hip_atomic_compare_exchange_strong(p, &expect, want, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// reserve up to `want` entries at the tail of ring `c` (its four control words); returns how many (0: no space)
RT_DEV int q_reserve_push(unsigned *c, int want, unsigned &start, int lane) {
    int got = 0;
    unsigned s = 0;
    if (lane == 0) {
        unsigned t = q_ld(c + QC_T_RES);
        for (int tries = 0; tries < 16; tries++) {
            const int space = RT_QCAP - (int)(t - q_ld(c + QC_H_DONE));
            const int take = want < space ? want : space;
            if (take <= 0) break;
            unsigned expect = t;
            if (q_cas(c + QC_T_RES, expect, t + (unsigned)take)) { got = take; s = t; break; }
            t = expect;
        }
    }
    start = (unsigned)__builtin_amdgcn_readfirstlane((int)s);
    return __builtin_amdgcn_readfirstlane(got);
}
// reserve up to `want` published entries at the head; none unless at least `min_take` are there
RT_DEV int q_reserve_pop(unsigned *c, int want, int min_take, unsigned &start, int lane) {
    int got = 0;
    unsigned s = 0;
    if (lane == 0) {
        unsigned h = q_ld(c + QC_H_RES);
        for (int tries = 0; tries < 16; tries++) {
            const int avail = (int)(q_ld(c + QC_T_DONE) - h);
            const int take = want < avail ? want : avail;
            if (take <= 0 || take < min_take) break;
            unsigned expect = h;
            if (q_cas(c + QC_H_RES, expect, h + (unsigned)take)) { got = take; s = h; break; }
            h = expect;
        }
    }
    start = (unsigned)__builtin_amdgcn_readfirstlane((int)s);
    return __builtin_amdgcn_readfirstlane(got);
}
// publish / release [start, start + n) once everything before `start` is in; false: gave up waiting (a bug: the kernel ends)
RT_DEV bool q_commit(unsigned *done, unsigned start, int n, int lane) {
    int ok = 1;
    if (lane == 0) {
        int spins = 0;
        while (q_ld(done) != start) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 21)) { ok = 0; break; }
        }
        if (ok) q_st(done, start + (unsigned)n);
    }
    return __builtin_amdgcn_readfirstlane(ok) != 0;
}

template <bool ALL_LDS, bool TAB_LDS, bool GEN_MEDIUM, int BLOCK>
__global__ void __launch_bounds__(BLOCK) render_kernel_q(SceneT<float> sc_arg, CameraT<float> cam_arg, RenderArgs a_arg, void *__restrict__ out_,
                                                         unsigned long long *__restrict__ counters) {
    using R = float;
    using LL = LdsLayout<float>;
    using Cell = unsigned long long;
    static_assert(BLOCK == LL::block && LL::tile_h == 1 && LL::item_slots == 8 && LL::waves <= 16 && LL::nan_words == 1, "entry ids: 3 + 3 + 4 bits");
    static_assert((RT_QCAP & (RT_QCAP - 1)) == 0 && RT_QCAP >= 128, "ring size");
#ifndef RT_Q_UNSAFE // (measurement builds only)
    static_assert(RT_Q_POP_MAX < 2 * RT_QCAP + 64 * (LL::waves - RT_Q_SHADERS), "the wait for space in the ray ring must end (see the shading trip)");
#endif
    float *const out = reinterpret_cast<float *>(out_);
    long long *const out64 = reinterpret_cast<long long *>(out_);
    unsigned *const hq = reinterpret_cast<unsigned *>(rt_smem + LL::queue_off);
    unsigned *const rq = reinterpret_cast<unsigned *>(rt_smem + LL::queue_off + LL::q_ring_bytes);
    unsigned *const ctl = reinterpret_cast<unsigned *>(rt_smem + LL::queue_off + 2 * LL::q_ring_bytes);
    if (threadIdx.x < 16) ctl[threadIdx.x] = 0u;
    SceneV<R, TAB_LDS> sc = stage_scene<R, TAB_LDS>(sc_arg); // (ends in a barrier)
    CameraT<R> cam = cam_arg;
    const RenderArgs &a = a_arg;
#define RT_OWN(x) asm volatile("" : "+s"(x))
    RT_OWN(sc.cold); RT_OWN(sc.image_bytes); RT_OWN(sc.tab); RT_OWN(sc.o_xforms); RT_OWN(sc.o_mats); RT_OWN(sc.o_texs);
    RT_OWN(sc.o_images); RT_OWN(sc.o_perlins); RT_OWN(sc.o_lights); RT_OWN(sc.n_lights); RT_OWN(sc.lights_is_list);
    RT_OWN(cam.W); RT_OWN(cam.H); RT_OWN(cam.sqrt_spp); RT_OWN(cam.max_depth); RT_OWN(cam.recip_sqrt_spp); RT_OWN(cam.defocus_angle);
    for (int k = 0; k < 3; k++) {
        RT_OWN(cam.center[k]); RT_OWN(cam.pixel00[k]); RT_OWN(cam.du[k]); RT_OWN(cam.dv[k]); RT_OWN(cam.ddu[k]); RT_OWN(cam.ddv[k]);
        RT_OWN(cam.background[k]);
    }
#undef RT_OWN
    const Mem<R, ALL_LDS> mem{sc_arg.stream, sc_arg.lds_units * Pos<R>::unit};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const R tmin = R(0.001);
    constexpr int NS = LL::item_slots, TPX = 8, TSH = 3;
    constexpr int WS_CNT = 4 * NS + 16 + NS; // paths of the slot's item that are still in the workgroup
    auto acc_of = [&](int w) { return reinterpret_cast<Cell *>(rt_smem + LL::wave_off + w * LL::wave_stride); };
    auto ws_of = [&](int w) { return reinterpret_cast<int *>(rt_smem + LL::wave_off + w * LL::wave_stride + LL::wave_acc * 8); };
    Cell *const acc = acc_of(wave);
    int *const ws = ws_of(wave);
    unsigned rays = 0;
    for (int k = lane; k < LL::wave_acc; k += 64) acc[k] = Cell(0);
    for (int k = lane; k < LL::state_ints; k += 64) ws[k] = 0;
    auto wsl = [&](int k) { return __builtin_amdgcn_readfirstlane(ws[k]); };
    int units_left = 0;

    // what a lane carries through the traversal of its ray
    Trav<R> tv;
    tv.i = 0;
    tv.hdr = K_IDLE;
    R tm = R(0);
    V3<R> beta = mk<R>(R(1), R(1), R(1));
    R med_xi = R(0);
    int smp = 0;
    unsigned ppb = 0; // pixel of the tile (3 bits) | slot (3) | wave (4) | bounce << 10
    auto pixel_of = [&](unsigned id) {
        const int *wo = ws_of((int)(id >> 6) & 15);
        const int slot = (int)(id >> 3) & 7;
        return (uint32_t)(__mul24(wo[4 * slot + WS_SLOT_J0], cam.W) + wo[4 * slot + WS_SLOT_I0] + (int)(id & 7u));
    };

    auto flush = [&](int slot, int i0, int j0, int chunk) {
        const int cell0 = (lane & (TPX - 1)) * 3;
        Cell *const b = acc + slot * LL::slot_acc + cell0;
        const int i = i0 + (lane & 7), j = j0;
        if (lane < TPX && i < cam.W && j < cam.H) {
            const size_t o = ((size_t)chunk * cam.H * cam.W + (size_t)j * cam.W + i) * 3;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const int bit = cell0 + c;
                const bool nan = (ws[WS_NAN + slot] >> bit) & 1;
                if (a.out_i64) out64[o + c] = nan ? I64Out::nan_cell : (long long)b[c];
                else out[o + c] = nan ? __builtin_nanf("") : (float)((double)b[c] * (1.0 / 16777216.0));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        if (lane < TPX) { b[0] = Cell(0); b[1] = Cell(0); b[2] = Cell(0); }
        if (lane == 0) ws[WS_NAN + slot] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    };
    // a finished path: its sample goes to its item's accumulator, wherever that is
    auto add_sample = [&](unsigned id, const V3<R> &L) {
        const int w = (int)(id >> 6) & 15, pp = (int)(id & 63u);
        Cell *const px = acc_of(w) + pp * 3;
        int *const wo = ws_of(w);
        const float lv[3] = {L.x, L.y, L.z};
        const float sum = (L.x + L.y) + L.z;
        if (sum < 256.0f && fminf(fminf(L.x, L.y), L.z) >= 0.0f) {
#pragma unroll
            for (int c = 0; c < 3; c++) lds_add(px + c, (unsigned long long)(unsigned)fmaf(lv[c], 16777216.0f, 0.5f));
        } else {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                if (lv[c] != lv[c]) {
                    const int bit = (pp & (TPX - 1)) * 3 + c;
                    (void)__hip_atomic_fetch_or(&wo[WS_NAN + (pp >> TSH)], 1 << bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                } else {
                    lds_add(px + c, to_fixed(__builtin_amdgcn_fmed3f(lv[c], 0.0f, RT_FX_CLAMP)));
                }
            }
        }
        (void)__hip_atomic_fetch_add(&wo[WS_CNT + (pp >> TSH)], -1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto fld = [&](unsigned *ring, int f, unsigned slot) -> unsigned & { return ring[f * RT_QCAP + (slot & (RT_QCAP - 1))]; };
    auto fbits = [](float x) { return __builtin_bit_cast(unsigned, x); };
    auto bitsf = [](unsigned x) { return __builtin_bit_cast(float, x); };

    int naps = 0;
#ifdef RT_PROFILE_ACTIONS
    unsigned long long prof_cyc[A_COUNT] = {}, prof_trips[A_COUNT] = {}, prof_lanes[A_COUNT] = {};
    const unsigned long long t_wave0 = __builtin_amdgcn_s_memtime();
    unsigned long long prof_idle = 0, prof_ready = 0, prof_all = 0, prof_nh = 0, prof_nq = 0, prof_pop = 0, prof_room = 0, sh_t[5] = {};
#endif
    for (;;) {
        const int kind = tv.hdr & 15;
        const bool is_box = kind == K_BOX, is_sph = kind == K_SPHERE || kind == K_MSPHERE, is_aab = kind == K_AABOX, is_med = kind == K_MEDIUM;
        const bool is_oth = kind >= K_QUAD && kind < K_IDLE && kind != K_MSPHERE;
        const bool ready = kind == K_READY, idle = kind == K_IDLE;
        const unsigned long long mb = lane_mask(is_box), ms = lane_mask(is_sph), ma = lane_mask(is_aab), mm = lane_mask(is_med),
                                 mo = lane_mask(is_oth), mr = lane_mask(ready), mi = lane_mask(idle);
        const unsigned long long ml = ms | ma | mm | mo;
        const int n_trav = popc64(mb | ml), nr = popc64(mr), ni = popc64(mi);
        // the rings' state (a snapshot: the reservations below decide)
        auto peek = [&](int k) { return __builtin_amdgcn_readfirstlane((int)__hip_atomic_load(ctl + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)); };
        const int h_tres = peek(QC_HIT + QC_T_RES), h_tdone = peek(QC_HIT + QC_T_DONE), h_hres = peek(QC_HIT + QC_H_RES), h_hdone = peek(QC_HIT + QC_H_DONE);
        const int r_tdone = peek(QC_RAY + QC_T_DONE), r_hres = peek(QC_RAY + QC_H_RES);
        const int room = peek(QC_ROOM), pop = peek(QC_POP);
        if (peek(QC_PANIC)) break;
#ifdef RT_PROFILE_ACTIONS
        const unsigned long long t_a = __builtin_amdgcn_s_memtime();
#endif
        const int nh = h_tdone - h_hres, nq = r_tdone - r_hres, hit_space = RT_QCAP - (h_tres - h_hdone);
        const bool push_ok = nr > 0 && hit_space > 0, pop_ok = ni > 0 && nq > 0;
        const bool deal_ok = units_left >= 0 && pop <= RT_Q_POP_MAX - 64;

        int action = A_NONE;
        int n_hits = 0;
        unsigned h_start = 0;
        if (room < RT_Q_SHADERS && (nh >= RT_Q_SHADE_MIN || deal_ok || (n_trav == 0 && nh > 0 && !push_ok && !pop_ok))) {
            // at most RT_Q_SHADERS waves shade at a time (the others keep the ray ring moving: see the push below)
            int in = 0;
            if (lane == 0) {
                const int old = (int)q_add(ctl + QC_ROOM, 1u);
                if (old < RT_Q_SHADERS) in = 1;
                else (void)q_add(ctl + QC_ROOM, (unsigned)-1);
            }
            in = __builtin_amdgcn_readfirstlane(in);
            if (in) {
                // a full trip of hits; else (the workgroup holds fewer paths than it may) a trip that only deals 64 new units;
                // any number of hits only when this wave has nothing else to do
                if (nh >= RT_Q_SHADE_MIN) n_hits = q_reserve_pop(ctl + QC_HIT, 64, RT_Q_SHADE_MIN, h_start, lane);
                else if (!deal_ok) n_hits = q_reserve_pop(ctl + QC_HIT, 64, 1, h_start, lane);
                if (n_hits > 0 || deal_ok) action = A_SHADE;
                else if (lane == 0) (void)q_add(ctl + QC_ROOM, (unsigned)-1);
            }
        }
        if (action == A_NONE && ((nr >= RT_Q_XMIN && hit_space > 0) || (ni >= RT_Q_XMIN && nq > 0) || (n_trav == 0 && (push_ok || pop_ok))))
            action = A_XCHG;
        if (action == A_NONE && n_trav > 0) {
            const unsigned sb = ((unsigned)(popc64(mb) * RT_W_BOX) << 3) | A_BOX, sl = ((unsigned)(popc64(ml) * RT_W_LEAF) << 3) | A_LEAF;
            action = (int)(smax_u32(sb, sl) & 7u);
        }
        if (action == A_NONE) {
            if (units_left < 0 && pop == 0) break; // no path is left anywhere in the workgroup and this wave has no units to deal
            __builtin_amdgcn_s_sleep(8);
#ifdef RT_PROFILE_ACTIONS
            prof_cyc[A_NONE] += __builtin_amdgcn_s_memtime() - t_a; prof_trips[A_NONE] += 1;
#endif
            if (++naps > (1 << 20)) { if (lane == 0) q_st(ctl + QC_PANIC, 3u); break; } // (a bug: paths are counted but nowhere to be found)
            continue;
        }
        naps = 0;
        const bool leaf_trip = action == A_LEAF;

        auto medium_xi_of = [&](int node_id, int sel) {
            R x = med_xi;
            if (sel != 1) x = medium_draw<R>(a.k0, a.k1, pixel_of(ppb), (uint32_t)smp, node_id, (int)(ppb >> 10), sel >> 1);
            return x;
        };
        auto leaf_round = [&](const int kind, const bool is_sph, const bool is_aab, const bool is_med, const bool is_oth,
                              const unsigned long long ms, const unsigned long long ma, const unsigned long long mm,
                              const unsigned long long mo) {
            if (leaf_trip && ms != 0ull) {
                if (is_sph) {
                    V3<R> cc = mk<R>(tv.u0.w[0], tv.u0.w[1], tv.u0.w[2]);
                    int next = tv.i + 2 * Pos<R>::unit;
                    if (kind == K_MSPHERE) {
                        const UnitT<R> u2 = mem.ld(next);
                        cc = cc + scale<R>(tm, mk<R>(u2.w[0], u2.w[1], u2.w[2]));
                        next += Pos<R>::unit;
                    }
                    R t;
                    if (sphere_t(cc, tv.u0.w[3], tv.o, tv.d, tmin, tv.best_t, t)) { tv.best_t = t; tv.best_off = tv.i; }
                    trav_goto(mem, tv, next);
                }
            }
            if (leaf_trip && ma != 0ull) {
                if (is_aab) {
                    R t;
                    if (aabox_t(tv.u0, tv.u1, tv.o, tv.inv, tmin, tv.best_t, t)) { tv.best_t = t; tv.best_off = tv.i; }
                    trav_goto(mem, tv, tv.i + 2 * Pos<R>::unit);
                }
            }
            if (leaf_trip && mm != 0ull) {
                if (is_med) {
                    R t;
                    if (medium_t(mk<R>(tv.u0.w[0], tv.u0.w[1], tv.u0.w[2]), tv.u0.w[3], tv.u1.w[0], ldi(tv.u1.w[2]), hdr_payload(tv.hdr), tv.o, tv.d,
                                 tmin, tv.best_t, medium_xi_of, t)) {
                        tv.best_t = t; tv.best_off = tv.i;
                    }
                    trav_goto(mem, tv, tv.i + 2 * Pos<R>::unit);
                }
            }
            if (leaf_trip && mo != 0ull) {
                if (is_oth) {
                    const RayT<R> world{tv.o, tv.d, tm}; // (the float path reads the ray's time only)
                    trav_leaf_step<GEN_MEDIUM>(mem, sc, tv, world, tmin, medium_xi_of);
                }
            }
        };
        if (action == A_BOX) {
            bool at_box = is_box;
            for (int rep = 0; rep < RT_BOX_REPEAT; rep++) {
                if (at_box) trav_box_step(mem, tv, tmin);
                int h = tv.hdr;
                at_box = h == make_hdr(K_BOX, 0);
                if (popc64(lane_mask(at_box)) < RT_BOX_REPEAT_MIN) break;
            }
        }
        if (leaf_trip) {
            leaf_round(kind, is_sph, is_aab, is_med, is_oth, ms, ma, mm, mo);
            for (int round = 1; round < RT_LEAF_ROUNDS; round++) {
                const int k2 = tv.hdr & 15;
                const bool s2 = k2 == K_SPHERE || k2 == K_MSPHERE, a2 = k2 == K_AABOX, m2 = k2 == K_MEDIUM;
                const bool o2 = k2 >= K_QUAD && k2 < K_IDLE && k2 != K_MSPHERE;
                const unsigned long long ms2 = lane_mask(s2), ma2 = lane_mask(a2), mm2 = lane_mask(m2), mo2 = lane_mask(o2);
                if (popc64(ms2 | ma2 | mm2 | mo2) < RT_LEAF_AGAIN_MIN) break;
                leaf_round(k2, s2, a2, m2, o2, ms2, ma2, mm2, mo2);
            }
        }
        // The exchange trip: finished rays leave their lanes for the hit ring, idle lanes take their next ray from the ray ring.
        if (action == A_XCHG) {
            bool is_idle = idle;
            if (nr > 0) {
                unsigned start;
                const int got = q_reserve_push(ctl + QC_HIT, nr, start, lane);
                if (got > 0) {
                    const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mr >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mr, 0u));
                    if (ready && rank < got) {
                        const unsigned s = start + (unsigned)rank;
                        fld(hq, 0, s) = fbits(tv.o.x); fld(hq, 1, s) = fbits(tv.o.y); fld(hq, 2, s) = fbits(tv.o.z);
                        fld(hq, 3, s) = fbits(tv.d.x); fld(hq, 4, s) = fbits(tv.d.y); fld(hq, 5, s) = fbits(tv.d.z);
                        fld(hq, 6, s) = fbits(tm); fld(hq, 7, s) = fbits(tv.best_t); fld(hq, 8, s) = (unsigned)tv.best_off;
                        fld(hq, 9, s) = fbits(beta.x); fld(hq, 10, s) = fbits(beta.y); fld(hq, 11, s) = fbits(beta.z);
                        fld(hq, 12, s) = (unsigned)smp; fld(hq, 13, s) = ppb;
                        tv.hdr = K_IDLE;
                        is_idle = true;
                    }
                    if (!q_commit(ctl + QC_HIT + QC_T_DONE, start, got, lane)) { if (lane == 0) q_st(ctl + QC_PANIC, 1u); break; }
                }
            }
            const unsigned long long mi2 = lane_mask(is_idle);
            if (mi2) {
                unsigned start;
                const int got = q_reserve_pop(ctl + QC_RAY, popc64(mi2), 1, start, lane);
                if (got > 0) {
                    const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mi2 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mi2, 0u));
                    const bool take = is_idle && rank < got;
                    RayT<R> ray;
                    ray.o = mk<R>(R(0), R(0), R(0)); ray.d = mk<R>(R(1), R(0), R(0)); ray.tm = R(0);
                    if (take) {
                        const unsigned s = start + (unsigned)rank;
                        ray.o = mk<R>(bitsf(fld(rq, 0, s)), bitsf(fld(rq, 1, s)), bitsf(fld(rq, 2, s)));
                        ray.d = mk<R>(bitsf(fld(rq, 3, s)), bitsf(fld(rq, 4, s)), bitsf(fld(rq, 5, s)));
                        tm = bitsf(fld(rq, 6, s));
                        beta = mk<R>(bitsf(fld(rq, 7, s)), bitsf(fld(rq, 8, s)), bitsf(fld(rq, 9, s)));
                        med_xi = bitsf(fld(rq, 10, s));
                        smp = (int)fld(rq, 11, s);
                        ppb = fld(rq, 12, s);
                    }
                    if (!q_commit(ctl + QC_RAY + QC_H_DONE, start, got, lane)) { if (lane == 0) q_st(ctl + QC_PANIC, 1u); break; }
                    if (take) trav_begin(mem, tv, ray);
                }
            }
        }
        // The shading trip: 64 entries of the hit ring (fewer when the workgroup runs dry), one per lane; paths that end are
        // replaced by new units of this wave's item while the workgroup holds fewer than RT_Q_POP_MAX paths; every ray that
        // starts goes to the ray ring.  (64 room credits were taken above.)
        if (action == A_SHADE) {
#ifdef RT_PROFILE_ACTIONS
            unsigned long long t_s = __builtin_amdgcn_s_memtime();
#endif
            bool has = lane < n_hits;
            RayT<R> ray;
            ray.o = mk<R>(R(0), R(0), R(0)); ray.d = mk<R>(R(1), R(0), R(0)); ray.tm = R(0);
            V3<R> hbeta = mk<R>(R(1), R(1), R(1));
            R best_t = R(0);
            int best_off = -1, hsmp = 0;
            unsigned id = 0;
            if (has) {
                const unsigned s = h_start + (unsigned)lane;
                ray.o = mk<R>(bitsf(fld(hq, 0, s)), bitsf(fld(hq, 1, s)), bitsf(fld(hq, 2, s)));
                ray.d = mk<R>(bitsf(fld(hq, 3, s)), bitsf(fld(hq, 4, s)), bitsf(fld(hq, 5, s)));
                ray.tm = bitsf(fld(hq, 6, s));
                best_t = bitsf(fld(hq, 7, s));
                best_off = (int)fld(hq, 8, s);
                hbeta = mk<R>(bitsf(fld(hq, 9, s)), bitsf(fld(hq, 10, s)), bitsf(fld(hq, 11, s)));
                hsmp = (int)fld(hq, 12, s);
                id = fld(hq, 13, s);
            }
            if (n_hits > 0 && !q_commit(ctl + QC_HIT + QC_H_DONE, h_start, n_hits, lane)) { if (lane == 0) q_st(ctl + QC_PANIC, 1u); break; }

#ifdef RT_PROFILE_ACTIONS
            { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); sh_t[0] += t_ - t_s; t_s = t_; }
#endif
            // ---- 1: hit records; paths that end without a draw
            Hit<R> h;
            h.p = mk<R>(R(0), R(0), R(0));
            h.mat = 0;
            bool scatter = false, ended = false;
            if (has) {
                V3<R> Ls;
                scatter = shade_hit(mem, sc, cam, ray, best_t, best_off, hbeta, Ls, h);
                if (!scatter) { add_sample(id, Ls); ended = true; has = false; }
            }

#ifdef RT_PROFILE_ACTIONS
            { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); sh_t[1] += t_ - t_s; t_s = t_; }
#endif
            // ---- 2: new units for the free lanes
            bool fresh = false;
            int fi = 0, fj = 0;
            const unsigned long long mfree = lane_mask(!has);
            if (mfree && units_left >= 0) {
                const int n_free = popc64(mfree);
                int grant = 0;
                if (lane == 0) {
                    const int old = (int)q_add(ctl + QC_POP, (unsigned)n_free);
                    int g = RT_Q_POP_MAX - old;
                    g = g < 0 ? 0 : (g > n_free ? n_free : g);
                    if (g < n_free) (void)q_add(ctl + QC_POP, (unsigned)-(n_free - g));
                    grant = g;
                }
                grant = __builtin_amdgcn_readfirstlane(grant);
                int dealt = 0, dealt_slot = 0;
                if (grant > 0 && units_left == 0) {
                    // the current item is dealt: take a slot none of whose paths is left in the workgroup (first in ring order)
                    const bool zero = lane < NS && __hip_atomic_load(&ws[WS_CNT + (lane & (NS - 1))], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0;
                    const unsigned free_slots = (unsigned)lane_mask(zero) & ((1u << NS) - 1u);
                    const int first = wsl(WS_CUR_SLOT) + 1;
                    const unsigned rot = ((free_slots >> first) | (free_slots << (NS - first))) & ((1u << NS) - 1u);
                    if (rot) {
                        int t = first + __builtin_ctz(rot);
                        t = t >= NS ? t - NS : t;
                        if (wsl(4 * t + WS_SLOT_VALID) != 0) flush(t, wsl(4 * t + WS_SLOT_I0), wsl(4 * t + WS_SLOT_J0), wsl(4 * t + WS_SLOT_CHUNK));
                        int item = 0;
                        if (lane == 0) item = (int)atomicAdd(&counters[1], 1ull);
                        item = __builtin_amdgcn_readfirstlane(item);
                        const int n_tiles = a.tiles_x * a.tiles_y;
                        const bool got = item < n_tiles * a.n_chunks;
                        const int chunk = item / n_tiles, tile = item - chunk * n_tiles;
                        const int tj = tile / a.tiles_x, ti = tile - tj * a.tiles_x;
                        int s0, s1;
                        if (a.chunk_size > 0) {
                            s0 = a.s_begin + chunk * a.chunk_size;
                            s1 = s0 + a.chunk_size;
                            if (s1 > a.s_end) s1 = a.s_end;
                        } else {
                            s0 = a.chunk_start[got ? chunk : 0];
                            s1 = a.chunk_start[got ? chunk + 1 : 0];
                        }
                        const int n_new = got && cam.max_depth > 0 ? (s1 - s0) << TSH : 0;
                        if (lane == 0) {
                            ws[4 * t + WS_SLOT_I0] = ti * 8; ws[4 * t + WS_SLOT_J0] = tj; ws[4 * t + WS_SLOT_CHUNK] = chunk;
                            ws[4 * t + WS_SLOT_VALID] = got ? 1 : 0;
                            ws[WS_CUR_SLOT] = t; ws[WS_CUR_S0] = s0; ws[WS_N_UNITS] = n_new;
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        units_left = got ? n_new : -1;
                    }
                }
                if (grant > 0 && units_left > 0) {
                    const int cur_slot = wsl(WS_CUR_SLOT), cur_s0 = wsl(WS_CUR_S0), n_item_units = wsl(WS_N_UNITS);
                    const int cur_i0 = wsl(4 * cur_slot + WS_SLOT_I0), cur_j0 = wsl(4 * cur_slot + WS_SLOT_J0);
                    int next_unit = n_item_units - units_left;
                    const int frank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mfree >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mfree, 0u));
                    bool need = !has && frank < grant;
                    unsigned long long m = lane_mask(need);
                    while (m && next_unit < n_item_units) {
                        const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                        const int un = next_unit + rank;
                        if (need && un < n_item_units) {
                            const int p = un & (TPX - 1);
                            const int i = cur_i0 + p, j = cur_j0;
                            if (i < cam.W && j < cam.H) {
                                need = false;
                                fresh = true;
                                fi = i;
                                fj = j;
                                id = (unsigned)(p | (cur_slot << TSH) | (wave << 6));
                                hsmp = cur_s0 + (un >> TSH);
                            }
                        }
                        next_unit += popc64(m);
                        m = lane_mask(need);
                    }
                    units_left = next_unit < n_item_units ? n_item_units - next_unit : 0;
                    dealt = popc64(lane_mask(fresh));
                    dealt_slot = cur_slot;
                }
                if (lane == 0) {
                    if (dealt) (void)__hip_atomic_fetch_add(&ws[WS_CNT + dealt_slot], dealt, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (dealt < grant) (void)q_add(ctl + QC_POP, (unsigned)-(grant - dealt));
                }
            }

#ifdef RT_PROFILE_ACTIONS
            { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); sh_t[2] += t_ - t_s; t_s = t_; }
#endif
            // ---- 3: one generator block per lane that starts a ray
            bool has_turb = false;
            R turb = R(0);
            if constexpr (RT_WAVE_NOISE) {
                int noise_idx = -1;
                if (scatter) {
                    const MatT<R> m = sc.mat(h.mat);
                    if (m.kind == RT_MAT_LAMBERTIAN || m.kind == RT_MAT_ISOTROPIC) {
                        const TexT<R> t = sc.tex(m.tex);
                        if (t.kind == RT_TEX_NOISE) noise_idx = t.a;
                    }
                }
                unsigned long long mn = lane_mask(noise_idx >= 0);
                while (mn) {
                    const int src = __builtin_ctzll(mn);
                    mn &= mn - 1;
                    auto from = [&](float x) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), src)); };
                    const V3<float> q = mk<float>(from(h.p.x), from(h.p.y), from(h.p.z));
                    const float tw = perlin_turb_wave(sc.perlin(__builtin_amdgcn_readlane(noise_idx, src)), q, 7, lane);
                    if (lane == src) { turb = tw; has_turb = true; }
                }
            }
            R xi5 = R(0);
            if (scatter || fresh) {
                int bounce = fresh ? 0 : (int)(id >> 10);
                const uint32_t pixel = fresh ? (uint32_t)(fj * cam.W + fi) : pixel_of(id);
                Rng<R> g;
                g.k0 = a.k0;
                g.k1 = a.k1;
                g.begin_stream(pixel, (uint32_t)hsmp, fresh ? 0u : (0x80000000u | (uint32_t)bounce));
                xi5 = g.fifth();
                if (fresh) {
                    int sj = (int)((float)hsmp * (float)cam.recip_sqrt_spp);
                    int si = hsmp - sj * cam.sqrt_spp;
                    if (si < 0) { sj--; si += cam.sqrt_spp; }
                    if (si >= cam.sqrt_spp) { sj++; si -= cam.sqrt_spp; }
                    ray = get_ray(cam, fi, fj, si, sj, g);
                    hbeta = mk<R>(R(1), R(1), R(1));
                } else {
                    shade_scatter(sc, ray, h, hbeta, g, has_turb, turb);
                    bounce++;
                    if (bounce == cam.max_depth) {
                        add_sample(id, hbeta * mk<R>(R(0), R(0), R(0)));
                        scatter = false;
                        ended = true;
                    }
                }
                id = (id & 1023u) | ((unsigned)bounce << 10);
            }

#ifdef RT_PROFILE_ACTIONS
            { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); sh_t[3] += t_ - t_s; t_s = t_; }
#endif
            // ---- 4: the rays go to the ray ring
            const bool go = scatter || fresh;
            const unsigned long long mgo = lane_mask(go);
            const int n_go = popc64(mgo);
            if (n_go) {
                // Space in the ray ring is not reserved ahead: the wave waits for it.  That cannot last: at most RT_Q_SHADERS
                // waves are here, the others take rays out whenever they have idle lanes, and they get idle lanes as long as
                // the hit ring takes their finished rays -- both rings full, every other lane holding a finished ray and the
                // shaders waiting would be more paths than RT_Q_POP_MAX (static_assert above).
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mgo >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mgo, 0u));
                int done = 0, waits = 0;
                bool bad = false;
                while (done < n_go) {
                    unsigned start;
                    const int got = q_reserve_push(ctl + QC_RAY, n_go - done, start, lane);
                    if (got == 0) {
                        __builtin_amdgcn_s_sleep(4);
                        if (++waits > (1 << 20)) { bad = true; break; }
                        continue;
                    }
                    if (go && rank >= done && rank < done + got) {
                        const unsigned s = start + (unsigned)(rank - done);
                        fld(rq, 0, s) = fbits(ray.o.x); fld(rq, 1, s) = fbits(ray.o.y); fld(rq, 2, s) = fbits(ray.o.z);
                        fld(rq, 3, s) = fbits(ray.d.x); fld(rq, 4, s) = fbits(ray.d.y); fld(rq, 5, s) = fbits(ray.d.z);
                        fld(rq, 6, s) = fbits(ray.tm);
                        fld(rq, 7, s) = fbits(hbeta.x); fld(rq, 8, s) = fbits(hbeta.y); fld(rq, 9, s) = fbits(hbeta.z);
                        fld(rq, 10, s) = fbits(xi5);
                        fld(rq, 11, s) = (unsigned)hsmp; fld(rq, 12, s) = id;
                    }
                    if (!q_commit(ctl + QC_RAY + QC_T_DONE, start, got, lane)) { bad = true; break; }
                    done += got;
                }
                if (bad) { if (lane == 0) q_st(ctl + QC_PANIC, 2u); break; }
                if (go) rays++;
            }

#ifdef RT_PROFILE_ACTIONS
            { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); sh_t[4] += t_ - t_s; t_s = t_; }
#endif
            const int n_end = popc64(lane_mask(ended));
            if (lane == 0) {
                (void)q_add(ctl + QC_ROOM, (unsigned)-1); // one shader fewer
                if (n_end) (void)q_add(ctl + QC_POP, (unsigned)-n_end);
            }
#ifdef RT_PROFILE_ACTIONS
            prof_lanes[A_SHADE] += (unsigned long long)(n_hits + popc64(lane_mask(fresh)));
#endif
        }
#ifdef RT_PROFILE_ACTIONS
        {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            const unsigned long long dt = __builtin_amdgcn_s_memtime() - t_a;
            prof_idle += ni; prof_ready += nr; prof_all += 1;
            prof_nh += nh; prof_nq += nq; prof_pop += pop; prof_room += room;
            const int n_act = action == A_BOX ? popc64(mb) : action == A_LEAF ? popc64(ml) : action == A_XCHG ? nr + ni : 0;
#pragma unroll
            for (int k = 0; k < A_COUNT; k++)
                if (action == k) { prof_cyc[k] += dt; prof_trips[k] += 1; prof_lanes[k] += (unsigned long long)n_act; }
        }
#endif
    }
#ifdef RT_PROFILE_ACTIONS
    if (lane == 0) {
        for (int k = 0; k < A_COUNT; k++) {
            atomicAdd(&counters[4 + 3 * k], prof_cyc[k]);
            atomicAdd(&counters[5 + 3 * k], prof_trips[k]);
            atomicAdd(&counters[6 + 3 * k], prof_lanes[k]);
        }
        atomicAdd(&counters[3], __builtin_amdgcn_s_memtime() - t_wave0);
        atomicAdd(&counters[33], prof_idle);
        atomicAdd(&counters[34], prof_ready);
        atomicAdd(&counters[35], prof_all);
        atomicAdd(&counters[22], prof_nh);
        atomicAdd(&counters[23], prof_nq);
        atomicAdd(&counters[24], prof_pop);
        atomicAdd(&counters[25], prof_room);
        for (int k = 0; k < 5; k++) atomicAdd(&counters[26 + k], sh_t[k]);
    }
#endif
    for (int k = 0; k < NS; k++)
        if (wsl(4 * k + WS_SLOT_VALID)) flush(k, wsl(4 * k + WS_SLOT_I0), wsl(4 * k + WS_SLOT_J0), wsl(4 * k + WS_SLOT_CHUNK));
    for (int off = 32; off > 0; off >>= 1) rays += __shfl_xor(rays, off);
    if (lane == 0) {
        atomicAdd(&counters[0], (unsigned long long)rays);
        const unsigned panic = q_ld(ctl + QC_PANIC);
        if (panic) atomicMax(&counters[2], (unsigned long long)panic);
    }
}

#endif
