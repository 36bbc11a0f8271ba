// One step of a pixel's walk: the ray in (o, d) with path weight w at tree node `node` is
// traced, shaded and replaced by its reflection child or by the ray on top of the stack; when
// nothing is left the step ends with RT_WALK_FINISHED.  This text is the body of the loops in
// whitted_exact.inc.h (included there twice: once per way a wave gets its pixels) and uses
// their variables; it is not a header of its own.
        RT_MARK("iter");
        RT_ENTER(cnt[0]);
#ifdef RT_PHASE_PROFILE
        ++cnt[7];  // lane-iterations: against 64 x cnt[0] this is the share of lanes still walking
#endif
        const int level = 32 - __builtin_clz(node);  // the reference's `depth`
        if (MW > 0) set_bit<MW>(smask, node);
        RT_EVENT(++c_rays);

        // ---- nearest hit, src/main.cc:37-47 ----
        int cur = -1;
        double nearest = __builtin_inf();
        find_nearest<COUNT, TREE, GRID>(a, geom, fgeom, o, d, node == 1u, nearest, cur, c_sph, c_pln);
        RT_MARK("iter_tail");
        if (MW > 0 && level == 1) first_hit = cur;

        bool next_in_regs = false;
        if (cur < 0) {
            RT_MARK("miss");
            RT_ENTER(cnt[2]);
            // src/main.cc:49-53: the background, weighted by the path.
            cr = __builtin_fmaf(w, bgr.x, cr); cg = __builtin_fmaf(w, bgr.y, cg); cb = __builtin_fmaf(w, bgr.z, cb);
        } else {
            RT_MARK("hit_setup");
            RT_ENTER(cnt[3]);
            if (MW > 0) set_bit<MW>(hmask, node);
            RT_EVENT(++c_hits; if (level == 1) ++c_phits);

            // per-lane gather of the hit object's records
            const DGeom* gsrc = a.geom + cur;
            const DMat* msrc = a.mat + cur;
            const double gr2 = gsrc->r2;
            const V3 gp = mk(gsrc->p[0], gsrc->p[1], gsrc->p[2]);
            const V3 gq = mk(msrc->plane_normal[0], msrc->plane_normal[1], msrc->plane_normal[2]);
            const F3 kd = f3(msrc->kd[0], msrc->kd[1], msrc->kd[2]);

            // src/main.cc:55-65
            const V3 ip = o + d * nearest;
            // calcNormal: a plane's is a constant of the plane (worked out at upload), a
            // sphere's is normalised here - when any lane of the wave has hit a sphere
            const bool on_plane = __double2hiint(gr2) < 0;
            V3 n = gq;
            if (__any(!on_plane)) {
                const V3 ns = unit(on_plane ? gq : (ip - gp));
                n = on_plane ? gq : ns;
            }
            const double nd0 = dot(n, d);
            const bool flipped = nd0 > 0;  // src/main.cc:62-65
            if (flipped) n = -n;
            const double nd = flipped ? -nd0 : nd0;  // dot(n, d) of the flipped normal, the same bits
            const F3 nf = f3((float)n.x, (float)n.y, (float)n.z);

            // ---- Object::Shade, src/objects.h:20-65 ----
            float sr = 0.0f, sg = 0.0f, sb = 0.0f;
            FilterRay ffr;  // launches without the anchor grids: shared by the feelers of this hit
            if (!GRID) ffr = filter_origin(ip, a.scene_p1max);
            for (int li = 0; li < a.n_lights; ++li) {
                RT_MARK("feeler_dir");
                RT_ENTER(cnt[4]);
                const LightRec L = load_light(lights, li);
                float occl = 1.0f;
                F3 fd;  // feeler.dir in single precision
                int first;
                RT_EVENT(++c_feel);
                if (GRID) {
                    first = feeler_by_grid<COUNT, TREE>(a, geom, fgeom, ltab, li, ip, mk(L.pos[0], L.pos[1], L.pos[2]), cur,
                                                        __double2hiint(gr2) < 0, occl, fd, c_fex, c_pln);
                } else {
                    const V3 fdir = unit(mk(L.pos[0], L.pos[1], L.pos[2]) - ip);  // feeler.dir
                    RT_MARK("feeler_loop");
                    first = shadow_feeler<COUNT, TREE, LATE, false>(a, geom, fgeom, ltab, li, ffr, ip, fdir, cur,
                                                                    __double2hiint(gr2) < 0, occl, c_fex, c_pln);
                    fd = ffr.d;
                }
                RT_EVENT(if (first >= 0) ++c_occ);
                RT_MARK("feeler_diffuse");
                // src/objects.h:50-60 (light_dir is the same vector as feeler.dir).  The reference
                // divides by length(normal) * length(light_dir): both are 1 to a few 2^-53, so the
                // quotient is left out.
                const float dk = __builtin_fabsf(dotf(nf, fd)) * occl;
                sr = __builtin_fmaxf(__builtin_fmaf(kd.x * L.ci[0], dk, sr), 0.0f);
                sb = __builtin_fmaxf(__builtin_fmaf(kd.z * L.ci[2], dk, sb), 0.0f);
                sg = __builtin_fmaxf(__builtin_fmaf(kd.y * L.ci[1], dk, sg), 0.0f);
            }

            // The material values the rest needs are fetched only now: registers are scarce
            // while the shadow loops run.  refraction_index, or 1.0 / refraction_index after
            // a flip (src/main.cc:64).
            __asm__ volatile("" : "+v"(msrc));
            const float m_refl = msrc->reflectivity;
            const float m_tr = msrc->transparency;
            const float m_amb = msrc->ambient_intensity;
            const double ri = flipped ? msrc->inv_refraction_index : msrc->refraction_index;

            // src/main.cc:70, 100-105
            RT_MARK("combine");
            // A child under a material coefficient of exactly zero has path weight zero and
            // adds nothing to the pixel, whatever it hits (the reference traces it and
            // multiplies by zero, src/main.cc:107-112): it is left out when only the image is
            // asked for, or when the caller asks for the ray tree that is actually walked
            // (KArgs::prune).
            const bool prune = (MW == 0 && !COUNT) || a.prune;
            const bool want_t = !(prune && m_tr == 0.0f);
            const bool want_r = !(prune && m_refl == 0.0f);
            V3 vt = mk(0.0, 0.0, 0.0);
            if (__any(want_t | want_r)) vt = div3(d, __builtin_fabs(nd));  // dot(d, n): the same products and sums
            const float g1 = 1.0f + (float)nd;  // 1 - facing, facing = -dot(n, d)
            const float fresnel = mix_tenth(g1 * g1 * g1, 1.0f);  // pow(1 - facing, 3), src/main.cc:101
            const float kt = (1.0f - fresnel) * m_tr;
            const float kr = (1.0f - kt) * m_refl;
            const float kdm = 1.0f - m_tr;
            const float amb = m_amb * a.inv_depth[level];

            // own colour: surface*kd + (bgr*ambient)*kd*(1/depth), src/main.cc:111-112
            const float wk = w * kdm;
            cr = __builtin_fmaf(wk, __builtin_fmaf(bgr.x, amb, sr), cr);
            cg = __builtin_fmaf(wk, __builtin_fmaf(bgr.y, amb, sg), cg);
            cb = __builtin_fmaf(wk, __builtin_fmaf(bgr.z, amb, sb), cb);

            // refraction test, src/main.cc:86-87 (asked only where a refraction child matters)
            double kk = 0.0;
            if (__any(want_t)) kk = (ri * ri) * len2(vt) - len2(vt + n);
            const float wr = w * kr, wt = w * kt;
            const bool go_t = want_t && kk > 0;
            const bool go_r = want_r;
            RT_EVENT(if (go_t) ++c_refr; if (want_t) ++c_ktest; if (go_r && level + 1 < a.max_depth) ++c_refl);

            if (level + 1 >= a.max_depth) {
                RT_MARK("children_last");
                // both children stop at the depth limit and return the background
                const float ws = go_t ? (wr + wt) : wr;  // (a child that is left out has weight zero)
                cr = __builtin_fmaf(ws, bgr.x, cr); cg = __builtin_fmaf(ws, bgr.y, cg); cb = __builtin_fmaf(ws, bgr.z, cb);
            } else {
                RT_MARK("children");
                RT_ENTER(cnt[6]);
                if (go_t) {
                    RT_MARK("child_refraction");
                    // src/main.cc:88-93; Ray's constructor normalises again (src/common.h:24)
                    V3 td = (n + vt) * inv_sqrt_2r(kk) - n;
                    td = unit2(td);
                    V3 to = ip - n * 1e-4;
                    // the weight (float) and the node number share the last field
                    const double ent[kStackFields] = {to.x, to.y, to.z, td.x, td.y, td.z,
                                                      __hiloint2double((int)(2 * node + 1), __float_as_int(wt))};
                    if (RT_STACK_IN_LDS) {  // (a constant of the including function)
                        if (sp < kStackLdsEntries) {
                            double* e = stk + (size_t)sp * kStackFields * 64;
#pragma unroll
                            for (int f = 0; f < kStackFields; ++f) e[f * 64] = ent[f];
                        } else {
#pragma unroll
                            for (int f = 0; f < kStackFields; ++f) last_entry[f] = ent[f];
                        }
                    } else if (sp < kStackRegEntries) {
#pragma unroll
                        for (int f = 0; f < kStackFields; ++f) bottom[f] = ent[f];
                    } else {
                        if (!DEEP || sp < kStackRegEntries + kStackLdsEntries) {
                            double* e = stk + (size_t)(sp - kStackRegEntries) * kStackFields * 64;
#pragma unroll
                            for (int f = 0; f < kStackFields; ++f) e[f * 64] = ent[f];
                        } else {
                            double* e = gstk + (size_t)(sp - kStackRegEntries - kStackLdsEntries) * kStackFields * 64 + lane;
#pragma unroll
                            for (int f = 0; f < kStackFields; ++f) e[f * 64] = ent[f];
                        }
                    }
                    ++sp;
                }
                // src/main.cc:75-79
                RT_MARK("child_reflection");
                if (go_r) {
                    V3 rd = unit2(vt + n * 2.0);
                    o = ip + n * 1e-4;
                    d = rd;
                    w = wr;
                    node = 2 * node;
                    next_in_regs = true;
                }
            }
        }

        RT_MARK("pop");
        if (!next_in_regs) {
            if (sp > RT_STACK_LOW) {  // (launches that share rays hand entries out from the low end)
                --sp;
                double ent[kStackFields];
                if (RT_STACK_IN_LDS) {
                    if (sp < kStackLdsEntries) {
                        const double* e = stk + (size_t)sp * kStackFields * 64;
#pragma unroll
                        for (int f = 0; f < kStackFields; ++f) ent[f] = e[f * 64];
                    } else {
#pragma unroll
                        for (int f = 0; f < kStackFields; ++f) ent[f] = last_entry[f];
                    }
                } else if (sp < kStackRegEntries) {
#pragma unroll
                    for (int f = 0; f < kStackFields; ++f) ent[f] = bottom[f];
                } else {
                    if (!DEEP || sp < kStackRegEntries + kStackLdsEntries) {
                        const double* e = stk + (size_t)(sp - kStackRegEntries) * kStackFields * 64;
#pragma unroll
                        for (int f = 0; f < kStackFields; ++f) ent[f] = e[f * 64];
                    } else {
                        const double* e = gstk + (size_t)(sp - kStackRegEntries - kStackLdsEntries) * kStackFields * 64 + lane;
#pragma unroll
                        for (int f = 0; f < kStackFields; ++f) ent[f] = e[f * 64];
                    }
                }
                o = mk(ent[0], ent[1], ent[2]);
                d = mk(ent[3], ent[4], ent[5]);
                w = __int_as_float(__double2loint(ent[6]));
                node = (uint32_t)__double2hiint(ent[6]);
            } else {
                RT_WALK_FINISHED;
            }
        }
