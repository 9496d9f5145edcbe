// codegen_group.cpp -- HIP emitter of the sixteen- and four-lanes-per-instance scheduled transient kernels
// (plan and rationale: group_plan.hpp).  The emitted kernel has the parameters and the hand-over
// protocol of the lane-per-instance kernel (codegen.cpp): fallback[], done[], violFlag.
#include "group_plan.hpp"

#include <cmath>
#include <cstdio>
#include <sstream>

namespace csim {

namespace {

std::string lit(double x)
{
    char buf[64];
    std::snprintf(buf, sizeof buf, "%a", x);
    return std::string("(") + buf + ")";
}

std::string intArray(const std::string& name, const std::vector<int32_t>& v)
{
    std::ostringstream o;
    o << "static __device__ const int " << name << "[" << (v.empty() ? 1 : v.size()) << "] = {";
    if (v.empty()) o << "0";
    for (std::size_t i = 0; i < v.size(); ++i) o << (i ? "," : "") << ((i % 32 == 31) ? "\n    " : "") << v[i];
    o << "};\n";
    return o.str();
}

// device code that does not depend on the circuit: element terms restated from the reference
// (the general kernels' versions live in device_common.hpp; a generated library is self-contained)
const char* kPrelude = R"GRP(
// ================= sixteen lanes per instance: shared device code =================
#define GRP_LANES 16
// broadcast of one lane's double to its DPP row of 16 lanes (v_mov_b64_dpp row_newbcast)
template <int L> __device__ __forceinline__ double grp_bc(double v)
{
    // bound_ctrl set: every lane has a source under row_newbcast, and the compiler then needs no
    // initialising move for the destination (one v_mov_b64_dpp per broadcast)
    return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + L, 0xF, 0xF, true);
}
// sum over the 16 lanes of a row, result in every lane
__device__ __forceinline__ double grp_sum16(double v)
{
    v += __builtin_amdgcn_update_dpp(0.0, v, 0x111, 0xF, 0xF, true);    // row_shr:1, zeros shifted in
    v += __builtin_amdgcn_update_dpp(0.0, v, 0x112, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0.0, v, 0x114, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0.0, v, 0x118, 0xF, 0xF, true);
    return grp_bc<15>(v);
}
// four lanes per instance: broadcast within a DPP quad (quad_perm [L,L,L,L]; two v_mov_b32_dpp -- the 64-bit DPP move
// knows row_newbcast only) and the sum over a quad, result in every lane
template <int L> __device__ __forceinline__ double grp_bc4(double v)
{
    return __builtin_amdgcn_update_dpp(0.0, v, L * 0x55, 0xF, 0xF, true);
}
__device__ __forceinline__ double grp_sum4(double v)
{
    v += __builtin_amdgcn_update_dpp(0.0, v, 0xB1, 0xF, 0xF, true);     // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0.0, v, 0x4E, 0xF, 0xF, true);     // quad_perm [2,3,0,1]
    return v;
}
// 1/a: v_rcp_f64 (measured: 2^-24.4 relative) and ONE cubic step r (1 + e + e^2), e = 1 - a r -- three dependent
// FMAs where two quadratic steps take four; on 2^20 random operands both give the correctly rounded reciprocal,
// bit for bit the same (tools/dev/ubench/rcp_acc.hip)
__device__ __forceinline__ double grp_rcp_nr(double a)
{
    const double r = __builtin_amdgcn_rcp(a);
    const double e = fma(-a, r, 1.0);
    return fma(fma(e, e, e), r, r);
}
// one wave per workgroup: LDS traffic of the wave is ordered by issue; this is the compiler fence
__device__ __forceinline__ void grp_sync() { __syncthreads(); }

// ordered sum of signed terms of one cell (plan.hpp: con = (term << 1) | negate)
__device__ __forceinline__ double grp_gather(const double* T, const int* ptr, const int* con, int cell)
{
    double acc = 0.0;
    for (int c = ptr[cell]; c < ptr[cell + 1]; ++c) {
        const int k = con[c];
        const double v = T[k >> 1];
        acc = (k & 1) ? acc - v : acc + v;
    }
    return acc;
}

#pragma clang fp contract(off)
__device__ __forceinline__ double grp_clamp01(double x) { return x < 0.0 ? 0.0 : (x > 1.0 ? 1.0 : x); }
// SourceSpec::evalTran (include/sim.hpp:160-162) with TranWaveform::eval (:75-143): SIN, PULSE, PWL
template <typename PGet>
__device__ __forceinline__ double grp_source_tran(PGet P, int wave, int waveN, double t, double pi)
{
    double w = 0.0;
    if (wave == GRP_WAVE_SIN) {
        const double v0 = P(1), va = P(2), freq = P(3), td = P(4), phi = P(5);
        if (t < td) w = v0;
        else { const double tau = t - td; const double om = 2.0 * pi * freq; w = v0 + va * sin(om * tau + phi); }
    } else if (wave == GRP_WAVE_PULSE) {
        const double v1 = P(1), v2 = P(2), td = P(3), tr = P(4), tf = P(5), ton = P(6), per = P(7);
        if (per <= 0.0) {
            const double tau = t - td;
            if (tau <= 0.0) w = v1;
            else if (tau < tr) { const double k = grp_clamp01(tau / tr); w = v1 + k * (v2 - v1); }
            else if (tau < tr + ton) w = v2;
            else { const double tfall = tau - (tr + ton); const double k = grp_clamp01(tfall / tf); w = v2 + k * (v1 - v2); }
        } else if (t < td) {
            w = v1;
        } else {
            double tau = fmod(t - td, per);
            if (tau < 0.0) tau += per;
            if (tau < tr) { const double k = grp_clamp01(tau / tr); w = v1 + (v2 - v1) * k; }
            else if (tau < tr + ton) w = v2;
            else if (tau < tr + ton + tf) { const double tfall = tau - (tr + ton); const double k = grp_clamp01(tfall / tf); w = v2 + (v1 - v2) * k; }
            else w = v1;
        }
    } else if (wave == GRP_WAVE_PWL) {
        const int n = waveN;
        if (n <= 0) w = 0.0;
        else if (t <= P(1)) w = P(1 + n);
        else if (t >= P(n)) w = P(2 * n);
        else {
            w = P(2 * n);
            for (int i = 0; i + 1 < n; ++i) {
                const double ta = P(1 + i), tb = P(2 + i);
                if (t > ta && t <= tb) {
                    const double va = P(1 + n + i), vb = P(2 + n + i);
                    const double k = (t - ta) / (tb - ta);
                    w = va + (vb - va) * k;
                    break;
                }
            }
        }
    }
    return P(0) + w;
}
// Level-1 MOSFET linearisation, MosfetBase::stamp (src/element.cpp:207-274); p = +1 (NMOS) / -1 (PMOS).
// Same operations as the reference, selects instead of branches, no contraction.
__device__ __forceinline__ void grp_mos_eval(double p, double Vth, double K, double lambda, double offGds,
                                             double Vd, double Vg, double Vs, double& gd, double& gg, double& gs, double& cst)
{
    const double Vgs = p * (Vg - Vs);
    const double Vds = p * (Vd - Vs);
    const double Vov = Vgs - Vth;
    const bool on = (Vgs > Vth) && (Vds >= 0.0);
    const bool tri = Vds < Vov;
    // both regions are evaluated and one is selected (v_cndmask): 16 lanes of a group may be in different
    // regions, and a branch would split the Newton loop's single basic block
    const double idTri = K * (Vov * Vds - 0.5 * Vds * Vds), idSat = 0.5 * K * Vov * Vov;
    const double gdsTri = K * (Vov - Vds);
    const double gmTri = K * Vds, gmSat = K * Vov;
    const double idOn = tri ? idTri : idSat, gdsOn = tri ? gdsTri : 0.0, gmOn = tri ? gmTri : gmSat;
    const double Ids0 = on ? idOn : 0.0;
    const double gds0 = on ? gdsOn : offGds;
    const double gm0 = on ? gmOn : 0.0;
    const double f1 = 1.0 + lambda * Vds;
    const double factor = f1 < 0.0 ? 0.0 : f1;
    const double Ids = p * (Ids0 * factor);
    gd = gds0 * factor + Ids0 * lambda;
    gg = gm0 * factor;
    gs = -(gd + gg);
    cst = Ids - gd * Vd - gg * Vg - gs * Vs;
}
#pragma clang fp contract(fast)
)GRP";

} // namespace

std::string groupPreludeSource(const csim_ir& ir)
{
    std::ostringstream o;
    o << "#define GRP_WAVE_SIN " << CSIM_WAVE_SIN << "\n#define GRP_WAVE_PULSE " << CSIM_WAVE_PULSE << "\n#define GRP_WAVE_PWL " << CSIM_WAVE_PWL << "\n"
      << "// outcome of one solve's pivot checks (see the first column of the elimination)\n"
      << "#define GRP_PIVOTS_BAD (worst > (0x1.0000000000008p+0) || tie >= 0.0 || pmin < " << lit(ir.k.lu_eps) << ")\n"
      << kPrelude << "\n";
    return o.str();
}

std::string emitGroupKernel(const csim_ir& ir, const AssemblyPlan& ap, const std::vector<PivotSchedule>& schedules,
                            const GeneratorOptions& gopt, GroupPlan* planOut, int lanes)
{
    if (schedules.empty()) return std::string();
    // The first (most frequent) schedule places the rows; the others are planned over that placement, so
    // that all solve bodies share the launch-constant matrix part, the staging rows and the scatter tables.
    std::vector<GroupPlan> plans(1);
    GroupPlan placed;
    const bool search = (gopt.placeSearch & (lanes == kGroupLanes ? 2 : 1)) != 0;
    if (search && !optimizeGroupPlacement(ir, ap, schedules, lanes, placed)) return std::string();
    if (!buildGroupPlan(ir, ap, schedules[0], plans[0], search ? &placed : nullptr, lanes)) return std::string();
    for (std::size_t a = 1; a < schedules.size(); ++a) {
        GroupPlan alt;
        if (!buildGroupPlan(ir, ap, schedules[a], alt, &plans[0], lanes)) return std::string();
        if (alt.gCellPtr != plans[0].gCellPtr || alt.gCellCon != plans[0].gCellCon || alt.iCellPtr != plans[0].iCellPtr ||
            alt.iCellCon != plans[0].iCellCon || alt.mosDest != plans[0].mosDest || alt.stageRows.size() != plans[0].stageRows.size() ||
            alt.gClasses.size() != plans[0].gClasses.size())
            return std::string();                              // cannot happen: the tables depend on the placement only
        plans.push_back(alt);
    }
    const GroupPlan& gp = plans[0];
    if (planOut) *planOut = gp;
    const bool multi = plans.size() > 1;
    const int N = gp.N, S = gp.S, G = gp.G;
    const bool quad = G != kGroupLanes;                        // four lanes per instance (DPP quads), 16 instances per wavefront
    // the small LDS image (see the carve-up below): always for four lanes per instance, and for sixteen from five rows
    // per lane on -- at N = 65 the full image is 90 KB per workgroup, one workgroup per CU
    const bool diet = quad || S > 4;
    const int perWave = 64 / G;
    const std::string GS = std::to_string(G), BC = quad ? "grp_bc4<" : "grp_bc<";
    const csim_consts& K = ir.k;
    const int NP = S * G;                                       // padded unknown count
    const int nMos = static_cast<int>(gp.mosElem.size());
    const int mosRounds = (nMos + G - 1) / G;
    const int nStage = static_cast<int>(gp.stageRows.size());
    // near-threshold guard (codegen.hpp GeneratorOptions::nearBand)
    const bool guard = gopt.nearBand > 0.0;

    const int nT1 = ap.nTerms + 1;                              // + one dummy term (always 0) for padded table entries

    std::ostringstream o;      // (the shared device code, groupPreludeSource(), is emitted by the caller)
    if (quad) o << "\nnamespace csim_q4 {   // (the tables below have the names of the 16-lane kernel's)\n";

    // ---- circuit tables
    {
        std::vector<int32_t> kind(ir.kind, ir.kind + ir.n_elems), eq(ir.eq, ir.eq + 4 * ir.n_elems),
            branch(ir.branch_eq, ir.branch_eq + ir.n_elems), slot(ir.param_slot, ir.param_slot + ir.n_elems),
            wave(ir.wave, ir.wave + ir.n_elems), waveN(ir.wave_n, ir.wave_n + ir.n_elems),
            tbase(ap.termBase.begin(), ap.termBase.end());
        o << intArray("grp_kind", kind) << intArray("grp_eq", eq) << intArray("grp_branch", branch)
          << intArray("grp_slot", slot) << intArray("grp_wave", wave) << intArray("grp_waveN", waveN)
          << intArray("grp_tbase", tbase)
          << intArray("grp_gPtr", gp.gCellPtr) << intArray("grp_gCon", gp.gCellCon)
          << intArray("grp_iPtr", gp.iCellPtr) << intArray("grp_iCon", gp.iCellCon);
        std::vector<int32_t> mosTab(static_cast<std::size_t>(std::max(1, mosRounds) * G), -1), destTab(static_cast<std::size_t>(std::max(1, mosRounds) * G * 8), -1);
        for (int m = 0; m < nMos; ++m) {
            mosTab[static_cast<std::size_t>(m)] = gp.mosElem[static_cast<std::size_t>(m)];
            for (int kd = 0; kd < 8; ++kd) destTab[static_cast<std::size_t>(m * 8 + kd)] = gp.mosDest[static_cast<std::size_t>(m)][static_cast<std::size_t>(kd)];
        }
        o << intArray("grp_mosElem", mosTab) << intArray("grp_mosDest", destTab);
    }
    // per-step work, lane-parallel with per-lane descriptors loaded once per launch:
    //  sources (round r: lane g evaluates source element grp_src[r*16+g]);
    //  history terms out = -TT[gterm] * (XP[a] - XP[b]) (tanalisis.cpp:77,308,337-341), packed a | b<<8, gterm | out<<16;
    //  right-hand side: per slot a padded list of signed per-step terms (index 2*term + negate into TS)
    std::vector<int32_t> srcTab;
    for (int e = 0; e < ir.n_elems; ++e) if (ir.kind[e] == CSIM_V || ir.kind[e] == CSIM_I) srcTab.push_back(e);
    const int srcRounds = (static_cast<int>(srcTab.size()) + G - 1) / G;
    srcTab.resize(static_cast<std::size_t>(std::max(1, srcRounds) * G), -1);
    // quad: per-step terms (a source's value, a history term) get compact slots in TS; sources' parameters are packed
    std::vector<int> stepSlot(static_cast<std::size_t>(nT1), -1);
    std::vector<int32_t> srcOff(srcTab.size(), 0), srcSlot(srcTab.size(), 0);
    int nSrcParams = 0, zeroSlot = 0;
    if (diet)
        for (std::size_t i = 0; i < srcTab.size(); ++i) {
            const int e = srcTab[i];
            if (e < 0) continue;
            srcOff[i] = nSrcParams;
            nSrcParams += (e + 1 < ir.n_elems ? ir.param_slot[e + 1] : ir.n_params) - ir.param_slot[e];
            stepSlot[static_cast<std::size_t>(ap.termBase[static_cast<std::size_t>(e)])] = zeroSlot;
            srcSlot[i] = zeroSlot++;
        }
    std::vector<int32_t> hA, hG;
    {
        const int dummy = ap.nTerms;
        // (quad: the history reads go through XH = XP - 1, whose entry 0 is the iterate's ground cell XS[NP])
        auto node = [&](int eq) { return diet ? (eq >= 0 ? eq + 1 : 0) : (eq >= 0 ? eq : NP); };
        auto add = [&](int gterm, int a, int b, int out) {
            if (diet) { stepSlot[static_cast<std::size_t>(out)] = zeroSlot; out = zeroSlot++; }
            hA.push_back(node(a) | (node(b) << 8));
            hG.push_back(gterm | (out << 16));
        };
        for (int e = 0; e < ir.n_elems; ++e) {
            const int tb = ap.termBase[static_cast<std::size_t>(e)];
            const int32_t* q = ir.eq + 4 * e;
            if (ir.kind[e] == CSIM_C) add(tb + T_C_GC, q[0], q[1], tb + T_C_IH);
            else if (ir.kind[e] == CSIM_L) { const int kb = ir.branch_eq[e]; add(tb + T_L_REQ, (kb >= 0 && kb < N) ? kb : -1, -1, tb + T_L_VH); }
            else if (ir.kind[e] == CSIM_NMOS || ir.kind[e] == CSIM_PMOS) {
                add(tb + T_M_GCH, q[1], q[2], tb + T_M_IHGS);
                add(tb + T_M_GCH, q[1], q[0], tb + T_M_IHGD);
                add(tb + T_M_GCF, q[2], q[3], tb + T_M_IHSB);
                add(tb + T_M_GCF, q[0], q[3], tb + T_M_IHDB);
            }
        }
        while (hA.size() % G) { hA.push_back(node(-1) | (node(-1) << 8)); hG.push_back(dummy | ((diet ? zeroSlot : dummy) << 16)); }
        if (ap.nTerms >= 65535 || NP >= 255) return std::string();
    }
    const int histRounds = static_cast<int>(hA.size()) / G;
    std::vector<int> rhsMax(static_cast<std::size_t>(S), 0);
    std::vector<int32_t> rhsIdx;
    for (int s = 0; s < S; ++s) {
        for (int lane = 0; lane < G; ++lane)
            rhsMax[static_cast<std::size_t>(s)] = std::max(rhsMax[static_cast<std::size_t>(s)],
                gp.iCellPtr[static_cast<std::size_t>(s * G + lane + 1)] - gp.iCellPtr[static_cast<std::size_t>(s * G + lane)]);
        for (int t = 0; t < rhsMax[static_cast<std::size_t>(s)]; ++t)
            for (int lane = 0; lane < G; ++lane) {
                const int lo = gp.iCellPtr[static_cast<std::size_t>(s * G + lane)], hi = gp.iCellPtr[static_cast<std::size_t>(s * G + lane + 1)];
                int con = lo + t < hi ? gp.iCellCon[static_cast<std::size_t>(lo + t)] : 2 * ap.nTerms;   // padding: + the dummy zero
                if (diet) {
                    const int sl = (con >> 1) == ap.nTerms ? zeroSlot : stepSlot[static_cast<std::size_t>(con >> 1)];
                    if (sl < 0) return std::string();          // a right-hand-side term that is not a per-step term: cannot happen
                    con = 2 * sl + (con & 1);
                }
                rhsIdx.push_back(con);
            }
    }
    o << intArray("grp_src", srcTab) << intArray("grp_hA", hA) << intArray("grp_hG", hG) << intArray("grp_rhs", rhsIdx);
    if (diet) o << intArray("grp_srcOff", srcOff) << intArray("grp_srcSlot", srcSlot);

    // LDS carve-up per instance (doubles)
    int oXS = 0, oXP = oXS + NP + 1, oTT = oXP + NP + 1, oTS = oTT + nT1, oPL = oTS + 2 * nT1, oST = oPL + ir.n_params;
    // Per-instance stride (generator option lds_pad): the four groups of a wave read the same [row][16] cells of their
    // own instance in one ds_read_b64, 32 lanes (two groups) per pass over the 64 four-byte banks; a stride of 16
    // doubles (mod 32) puts the second group of a pass on the other half of the banks.  Measured: no fewer conflicts
    // (codegen.hpp), so off by default.
    int instDoubles = oST + (nStage + 1) * G;
    if (diet) {
        // Four lanes per instance put 16 instances into a workgroup, and four workgroups must share a CU's 160 KB (one
        // wave per SIMD): 320 doubles per instance.  So: only the sources' parameters (PL), only the per-step terms in
        // TS (compact slots; zeroSlot = always 0), and TT -- needed while the launch-constant matrix part and the
        // history coefficients are gathered, dead afterwards -- shares its place with TS and the staging rows.
        // XP has no ground cell of its own (XH, above), and the staging rows' dummy is the one cell lanes without a
        // MOSFET scatter into.  The stride is then padded to 4 (mod 8) doubles: a 64-bit LDS access serves 16 lanes -- four
        // instances reading 8 consecutive dwords each -- per pass over the 64 banks, and with that stride they start
        // on different multiples of 8 dwords (measured: a stride of 320 doubles, all instances on the same banks, cost
        // 12 % of the kernel).
        oPL = oXP + NP;
        oTS = oPL + std::max(nSrcParams, 1);
        oTT = oTS;
        oST = oTS + 2 * (zeroSlot + 1);
        instDoubles = std::max(oST + nStage * G + (gopt.dummyOneCell ? 1 : G), oTT + nT1);
        while (quad && instDoubles % 8 != 4) ++instDoubles;
    }
    while (gopt.ldsPad && instDoubles % 32 != 16) ++instDoubles;
    if (instDoubles * 8 * perWave > 150 * 1024) return std::string();
    if (planOut) planOut->ldsDoubles = instDoubles;  // one CU's LDS (a workgroup of four instances must fit)


    o << (quad ? "\n// One DPP quad of 4 lanes = one circuit instance, 16 instances per wavefront (group_plan.hpp).\n"
               : "\n// One DPP row of 16 lanes = one circuit instance, 4 instances per wavefront (group_plan.hpp).\n")
      << "// pivot schedule" << (multi ? "s, tried in this order" : "") << ": " << [&] {
             std::string all;
             for (std::size_t a = 0; a < schedules.size(); ++a) all += (a ? " ; " : "") + (schedules[a].str().empty() ? std::string("-") : schedules[a].str());
             return all; }() << "\n"
      << "extern \"C\" __global__ void __launch_bounds__(64)"
      << (gopt.groupWavesPerEu > 0 ? " __attribute__((amdgpu_waves_per_eu(" + std::to_string(gopt.groupWavesPerEu) + ", " + std::to_string(gopt.groupWavesPerEu) + ")))" : std::string()) << "\n"
      << (quad ? "csim_tran_group4_kernel" : "csim_tran_group_kernel") << "(const double* __restrict__ params, int B, double dt, long long stepFirst,\n"
      << "                       long long nSteps, const int* __restrict__ probeEq, int nProbe, int outStride,\n"
      << "                       double* __restrict__ wave, double* __restrict__ xio, long long* __restrict__ iters,\n"
      << "                       unsigned* __restrict__ status, int* __restrict__ stepIters,\n"
      << "                       unsigned char* __restrict__ fallback, int* __restrict__ done,\n"
      << "                       int* __restrict__ violFlag, double* __restrict__ nearX, int* __restrict__ nearStep,\n"
      << "                       int* __restrict__ nearIt, long long* __restrict__ nearItAfter)\n{\n"
      << "    __shared__ double lds[" << perWave << " * " << instDoubles << "];\n"
      << "    const int lane = threadIdx.x, g = lane & " << G - 1 << ", q = lane >> " << (quad ? 2 : 4) << ";\n"
      << "    const int b = blockIdx.x * " << perWave << " + q;\n"
      << "    const bool inb = b < B;\n"
      << "    const long long bb = inb ? b : B - 1;      // out-of-range groups shadow the last instance, never store\n"
      << "    const long long SB = B;\n"
      << "    if (!__any(inb && done[bb] < nSteps)) return;\n"
      << "    double* const XS = lds + q * " << instDoubles << " + " << oXS << ";   // iterate; XS[" << NP << "] = 0 (ground)\n"
      << "    double* const XP = lds + q * " << instDoubles << " + " << oXP << ";   // state at the start of the step (history, checkpoint)\n"
      << "    double* const TT = lds + q * " << instDoubles << " + " << oTT << ";   // element terms (plan.hpp); TT[" << ap.nTerms << "] = 0 (padding)\n"
      << "    double* const TS = lds + q * " << instDoubles << " + " << oTS << ";   // per-step terms with sign: TS[2t] = +TT[t], TS[2t+1] = -TT[t]\n"
      << "    double* const PL = lds + q * " << instDoubles << " + " << oPL << ";   // this instance's parameters\n"
      << "    double* const ST = lds + q * " << instDoubles << " + " << oST << ";   // MOSFET staging rows [row][16]; last row = dummy\n"
      << (diet ? "    double* const XH = XP - 1;          // history view of XP: XH[0] = XS[" + std::to_string(NP) + "] = 0 (ground), XH[1 + i] = XP[i]\n" : "")
      << "    const unsigned long long rowBits = " << (quad ? "0xFull" : "0xFFFFull") << " << (" << G << " * q);\n"
      << "    auto P = [&](int slot) -> double { return params[(long long)slot * SB + bb]; };\n\n";

    // ---- launch setup: zero staging, constants, terms
    if (!diet)
        o << "    for (int i = g; i < " << (nStage + 1) * G << "; i += " << GS << ") ST[i] = 0.0;\n";
    o << "    for (int i = g; i < " << nT1 << "; i += " << GS << ") TT[i] = 0.0;\n";
    if (!diet)
        o << "    for (int i = g; i < " << 2 * nT1 << "; i += " << GS << ") TS[i] = 0.0;\n"
          << "    for (int i = g; i < " << ir.n_params << "; i += " << GS << ") PL[i] = P(i);\n";
    o
      << "    if (g == 0) { XS[" << NP << "] = 0.0; " << (diet ? std::string() : "XP[" + std::to_string(NP) + "] = 0.0; ") << "}\n"
      << "    grp_sync();\n"
      << "    bool badL = false;\n"
      << "    for (int e = g; e < " << ir.n_elems << "; e += " << GS << ") {          // terms constant over the launch (tanalisis.cpp:59-80,294-341)\n"
      << "        const int kind = grp_kind[e], s = grp_slot[e], tb = grp_tbase[e];\n"
      << "        if (kind == " << CSIM_R << ") { const double R = P(s); TT[tb + " << T_R_G << "] = (R == 0.0) ? 0.0 : 1.0 / R; }\n"
      << "        else if (kind == " << CSIM_C << ") { const double C = P(s); TT[tb + " << T_C_GC << "] = (C > 0.0 && dt > 0.0) ? C / dt : 0.0; }\n"
      << "        else if (kind == " << CSIM_L << ") { const double L = P(s); badL = badL || !(L > 0.0); TT[tb + " << T_L_REQ << "] = L / dt; TT[tb + " << T_L_ONE << "] = 1.0; }\n"
      << "        else if (kind == " << CSIM_NMOS << " || kind == " << CSIM_PMOS << ") {\n"
      << "            const double Cj0 = P(s + 3), Ch = 0.5 * Cj0;\n"
      << "            TT[tb + " << T_M_GCH << "] = (Ch > 0.0 && dt > 0.0) ? Ch / dt : 0.0;\n"
      << "            TT[tb + " << T_M_GCF << "] = (Cj0 > 0.0 && dt > 0.0) ? Cj0 / dt : 0.0;\n"
      << "        }\n"
      << "    }\n"
      << "    if (g == 0) { TT[" << ap.termOne << "] = 1.0; TT[" << ap.termGmin << "] = " << lit(K.tran_gmin) << "; }\n"
      << "    // the inductor incidence is folded as the exact constant 1: needs L > 0 (tanalisis.cpp:296), else general kernel\n"
      << "    bool viol = (__ballot(badL) & rowBits) != 0ull;\n"
      << "    grp_sync();\n\n";

    // per-lane 0/1 factors: rows of the slot being consumed that are still active at column k (g > k % 16)
    // (formed arithmetically so that they live in vector registers, not as scalar lane masks)
    for (int t = 0; t < G; ++t) o << "    const double mk" << t << " = fmin(fmax((double)(g - " << t << "), 0.0), 1.0);\n";
    // launch-constant part of the matrix
    o << "    // launch-constant part of every matrix class, terms summed in the reference's stamping order\n";
    std::vector<std::vector<int>> classOf(static_cast<std::size_t>(S), std::vector<int>(static_cast<std::size_t>(N + 1), -1));
    for (std::size_t c = 0; c < gp.gClasses.size(); ++c) {
        classOf[static_cast<std::size_t>(gp.gClasses[c].s)][static_cast<std::size_t>(gp.gClasses[c].j)] = static_cast<int>(c);
        o << "    const double c_" << gp.gClasses[c].s << "_" << gp.gClasses[c].j << " = grp_gather(TT, grp_gPtr, grp_gCon, " << c * G << " + g);\n";
    }
    // MOSFET lanes
    for (int r = 0; r < std::max(1, mosRounds); ++r) {
        const std::string R = std::to_string(r);
        o << "    // MOSFET evaluated by this lane in round " << r << " (-1: none)\n"
          << "    const int me" << R << " = grp_mosElem[" << r * G << " + g];\n"
          << "    const int mq" << R << " = me" << R << " >= 0 ? me" << R << " : 0;\n"
          << "    const int mD" << R << " = grp_eq[4 * mq" << R << "] >= 0 ? grp_eq[4 * mq" << R << "] : " << NP << ";\n"
          << "    const int mG" << R << " = grp_eq[4 * mq" << R << " + 1] >= 0 ? grp_eq[4 * mq" << R << " + 1] : " << NP << ";\n"
          << "    const int mS" << R << " = grp_eq[4 * mq" << R << " + 2] >= 0 ? grp_eq[4 * mq" << R << " + 2] : " << NP << ";\n"
          << "    const double mp" << R << " = grp_kind[mq" << R << "] == " << CSIM_PMOS << " ? -1.0 : 1.0;\n"
          << "    const double mvth" << R << " = P(grp_slot[mq" << R << "]), mK" << R << " = P(grp_slot[mq" << R << "] + 1), mlam" << R << " = P(grp_slot[mq" << R << "] + 2);\n";
        for (int kd = 0; kd < 8; ++kd)
            o << "    const int md" << R << "_" << kd << " = grp_mosDest[(" << r * G << " + g) * 8 + " << kd << "] >= 0 ? grp_mosDest[(" << r * G
              << " + g) * 8 + " << kd << "] : " << nStage * G << (gopt.dummyOneCell ? " + 0;\n" : " + g;\n");
    }

    for (int r = 0; r < srcRounds; ++r)
        o << "    const int se" << r << " = grp_src[" << r * G << " + g];       // source evaluated by this lane in round " << r << " (-1: none)\n"
          << "    const int sq" << r << " = se" << r << " >= 0 ? se" << r << " : 0;\n"
          << "    const int ssl" << r << " = grp_slot[sq" << r << "], stb" << r << (diet ? " = grp_srcSlot[" + std::to_string(r * G) + " + g]" : " = grp_tbase[sq" + std::to_string(r) + "]") << ", swv" << r << " = grp_wave[sq" << r
          << "], swn" << r << " = grp_waveN[sq" << r << "];\n";
    if (diet)
        for (int r = 0; r < srcRounds; ++r)
            o << "    const int spo" << r << " = grp_srcOff[" << r * G << " + g];\n"
              << "    if (se" << r << " >= 0) {\n"
              << "        const int np = (se" << r << " + 1 < " << ir.n_elems << " ? grp_slot[se" << r << " + 1] : " << ir.n_params << ") - ssl" << r << ";\n"
              << "        for (int i = 0; i < np; ++i) PL[spo" << r << " + i] = P(ssl" << r << " + i);\n"
              << "    }\n";
    for (int r = 0; r < histRounds; ++r)
        o << "    const int hA" << r << " = grp_hA[" << r * G << " + g], hG" << r << " = grp_hG[" << r * G << " + g];\n";
    if (diet) {
        for (int r = 0; r < histRounds; ++r) o << "    const double hc" << r << " = TT[hG" << r << " & 0xFFFF];\n";
        o << "    grp_sync();        // TT is dead from here on: its place is taken by the per-step terms and the staging rows\n"
          << "    for (int i = g; i < " << 2 * (zeroSlot + 1) << "; i += " << GS << ") TS[i] = 0.0;\n"
          << "    for (int i = g; i < " << nStage * G + (gopt.dummyOneCell ? 1 : G) << "; i += " << GS << ") ST[i] = 0.0;\n";
    }
    {
        int base = 0;
        for (int s = 0; s < S; ++s)
            for (int t = 0; t < rhsMax[static_cast<std::size_t>(s)]; ++t, ++base)
                o << "    const int ri" << s << "_" << t << " = grp_rhs[" << base * G << " + g];\n";
    }

    // ---- state
    o << "\n    // state: lane g keeps x[16 s + g]\n";
    for (int s = 0; s < S; ++s) {
        o << "    double xo" << s << " = (" << s * G << " + g < " << N << ") ? xio[(long long)(" << s * G << " + g) * SB + bb] : 0.0;\n"
          << "    XS[" << s * G << " + g] = xo" << s << "; XP[" << s * G << " + g] = xo" << s << ";\n";
    }
    o << "    grp_sync();\n"
      << "    unsigned st = inb ? status[bb] : 0u;\n"
      << "    bool dead = !inb || (st & ST_TRAN_NONFINITE) != 0u;   // the reference would have thrown: stay stopped\n"
      << "    long long itTotal = 0;\n"
      << "    long long sdone = (inb && !dead) ? (long long)done[bb] : nSteps;\n"
      << (guard ? "    int nearS = 0;          // step (of this launch) of the group's first near-threshold convergence decision; 0 = none\n" : "")
      << "    if (stepFirst == 0 && sdone == 0 && wave && inb)\n"
      << "        for (int pq = g; pq < nProbe; pq += " << GS << ") wave[((long long)pq) * SB + b] = XS[probeEq[pq]];\n";
    const bool piped = gopt.pipelineMos != 0 && mosRounds > 0 && nStage > 0;
    // ---- MOSFET evaluation + scatter into the staging rows
    auto emitMos = [&](const std::string& ind) {
        for (int r = 0; r < mosRounds; ++r) {
            const std::string R = std::to_string(r);
            o << ind << "{   // MOSFET channel at the iterate (element.cpp:207-274), lane m of the group evaluates MOSFET " << r * G << " + m\n"
              << ind << "    double gd, gg, gs, cst;\n"
              << ind << "    grp_mos_eval(mp" << R << ", mvth" << R << ", mK" << R << ", mlam" << R << ", " << lit(K.mos_off_gds)
              << ", vd" << R << ", vg" << R << ", vs" << R << ", gd, gg, gs, cst);\n"
              << ind << "    // lanes without a MOSFET write the dummy row (their destinations all point there): no branch\n"
              << ind << "    ST[md" << R << "_0] = gd; ST[md" << R << "_1] = gg; ST[md" << R << "_2] = gs; ST[md" << R << "_3] = -cst;\n"
              << ind << "    ST[md" << R << "_4] = -gd; ST[md" << R << "_5] = -gg; ST[md" << R << "_6] = -gs; ST[md" << R << "_7] = cst;\n"
              << ind << "}\n";
        }
    };
    if (piped) {
        // Software pipeline: the MOSFET pass for iteration i+1 and the reads of its staging rows run at the END of
        // iteration i (from the candidate iterate, under the convergence bookkeeping), so that at the head of an
        // iteration the staged values are already in registers -- the elimination used to wait there for a write ->
        // read round trip through the LDS.  The pass at the end of a step's last iteration saw the state the step
        // ended with (a group that is not iterating keeps its state in XS, and every pass re-evaluates it from
        // there), which is what the next step's first iteration needs: only the first step of a launch needs the
        // pass done for it, here.
        for (int r = 0; r < mosRounds; ++r)
            o << "    double vd" << r << " = XS[mD" << r << "], vg" << r << " = XS[mG" << r << "], vs" << r << " = XS[mS" << r << "];\n";
        emitMos("    ");
        for (int r = 0; r < nStage; ++r) o << "    double sv" << r << " = ST[" << r * G << " + g];\n";
    }
    o << "    int smin = (int)(sdone < nSteps ? sdone + 1 : nSteps + 1);\n"
      << "    for (int m = 32; m >= 1; m >>= 1) { const int ot = __shfl_xor(smin, m); smin = ot < smin ? ot : smin; }\n"
      << "    smin = __builtin_amdgcn_readfirstlane(smin);\n"
      << "    // output decimation without a 64-bit division per step: phase = gstep % outStride, orow = gstep / outStride\n"
      << "    int ophase = (int)((stepFirst + smin) % outStride);\n"
      << "    long long orow = (stepFirst + smin) / outStride;\n"
      << "    for (long long s = smin; s <= nSteps; ++s, ++ophase) {\n"
      << "        if (ophase == outStride) { ophase = 0; ++orow; }\n"
      << "        if (!__any(!dead && !viol && sdone < nSteps)) break;\n"
      << "        const bool live = !dead && !viol && sdone + 1 == s;\n"
      << "        const long long gstep = stepFirst + s;\n"
      << "        const double tNow = (double)(int)gstep * dt;\n"
      << "        if (live) {";
    for (int s = 0; s < S; ++s) o << " XP[" << s * G << " + g] = xo" << s << ";";
    o << " }\n"
      << "        // per-step terms: sources (sim.hpp:160-162) and history currents (tanalisis.cpp:77,308,337-341).  All LDS reads\n"
      << "        // of a phase are issued before its first write (the compiler must keep a read behind an earlier write of the\n"
      << "        // same array): the history operands and the first iteration's MOSFET inputs fly while the sources are\n"
      << "        // evaluated, the right-hand-side terms in one batch after the writes.\n";
    for (int r = 0; r < histRounds; ++r)
        o << "        const double " << (diet ? std::string() : "hc" + std::to_string(r) + " = TT[hG" + std::to_string(r) + " & 0xFFFF], ") << "hp" << r << " = " << (diet ? "XH" : "XP") << "[hA" << r << " & 0xFF], hq" << r << " = " << (diet ? "XH" : "XP") << "[hA" << r << " >> 8];\n";
    if (!piped)
        for (int r = 0; r < mosRounds; ++r)
            o << "        double vd" << r << " = XS[mD" << r << "], vg" << r << " = XS[mG" << r << "], vs" << r << " = XS[mS" << r << "];\n";
    for (int r = 0; r < srcRounds; ++r)
        o << "        if (se" << r << " >= 0) {\n"
          << "            const double v = grp_source_tran([&](int i) { return PL[" << (diet ? "spo" : "ssl") << r << " + i]; }, swv" << r << ", swn" << r << ", tNow, " << lit(K.pi) << ");\n"
          << "            TS[2 * stb" << r << "] = v; TS[2 * stb" << r << " + 1] = -v;\n"
          << "        }\n";
    for (int r = 0; r < histRounds; ++r)
        o << "        {\n"
          << "            const double v = -hc" << r << " * (hp" << r << " - hq" << r << ");\n"
          << "            TS[2 * (hG" << r << " >> 16)] = v; TS[2 * (hG" << r << " >> 16) + 1] = -v;\n"
          << "        }\n";
    o << "        // right-hand side without the MOSFET terms, per-step terms summed in the reference's stamping order\n";
    for (int s = 0; s < S; ++s)
        for (int t = 0; t < rhsMax[static_cast<std::size_t>(s)]; ++t) o << "        const double rt" << s << "_" << t << " = TS[ri" << s << "_" << t << "];\n";
    for (int s = 0; s < S; ++s) {
        o << "        double cb" << s << " = 0.0;\n";
        for (int t = 0; t < rhsMax[static_cast<std::size_t>(s)]; ++t) o << "        cb" << s << " += rt" << s << "_" << t << ";\n";
    }

    const int slowIters = slowStepIters(K.tran_tol, K.tran_alpha, K.tran_max_iters);
    const std::string in = "            ";
    o << "        bool active = live;\n"
      << "        int it = 0;\n"
      << (guard ? (gopt.nearForm == 0 ? "        double nearMin = 1.0;   // smallest |ss - tol^2| of this step's passes\n"
                                      : "        bool nearAny = false;   // some pass of this step had err within the band around tol\n") : "");
    o << "        for (int iter = 0; iter < " << K.tran_max_iters << "; ++iter) {\n"
      << "            if (!__any(active)) break;\n";
    if (!piped) emitMos(in);
    // No fence here or anywhere inside the Newton loop: the LDS executes one wave's instructions in issue
    // order, and the compiler keeps a store and a later load of the same LDS array in program order (they
    // may alias).  A __syncthreads() would add nothing but a scheduling barrier and a full lgkmcnt(0) wait
    // (measured: 30 % of the wave's cycles in s_waitcnt with four of them per iteration).

    // ---- one solve: assembly, elimination with the plan's pivots, back substitution into <xname><slot>
    auto maskName = [&](const GroupPlan::Column::SlotMask& m) -> std::string {
        if (m.keepAll) return std::string();
        if (m.suffix >= 0) return "mk" + std::to_string(m.suffix);
        char buf[48];
        std::snprintf(buf, sizeof buf, "(((0x%04x >> g) & 1) ? 1.0 : 0.0)", m.lanes & 0xFFFFu);
        return buf;
    };
    auto emitSolve = [&](const GroupPlan& pl, const std::string& in, const std::string& xname) -> bool {
    // ---- assembly: class registers
    o << in << "// assembly: (terms constant within the step) + (MOSFET terms from the staging rows, in stamping order)\n";
    std::vector<std::vector<char>> declared(static_cast<std::size_t>(S), std::vector<char>(static_cast<std::size_t>(N + 1), 0));
    for (int s = 0; s < S; ++s)
        for (int j = 0; j <= N; ++j) {
            if (!pl.classLive[static_cast<std::size_t>(s)][static_cast<std::size_t>(j)]) continue;
            declared[static_cast<std::size_t>(s)][static_cast<std::size_t>(j)] = 1;
            std::string init = "0.0";
            if (j == N) init = "cb" + std::to_string(s);
            else if (classOf[static_cast<std::size_t>(s)][static_cast<std::size_t>(j)] >= 0) init = "c_" + std::to_string(s) + "_" + std::to_string(j);
            o << in << "double a_" << s << "_" << j << " = " << init << ";\n";
        }
    for (int r = 0; r < nStage; ++r)
        if (!declared[static_cast<std::size_t>(pl.stageRows[static_cast<std::size_t>(r)].s)][static_cast<std::size_t>(pl.stageRows[static_cast<std::size_t>(r)].j)])
            return false;                                              // cannot happen: a staged cell is live
    // A staging row is read a few columns before the column that first touches its class and added right
    // there: issued up front, the reads end up one after the other, each waited for (the kernel sits at the
    // 256 architectural registers, so the compiler reuses one landing register: 13 LDS round trips, 28 % of
    // the wave's cycles); spread over the elimination they overlap with it.
    std::vector<int> firstUse(static_cast<std::size_t>(nStage), N);   // column whose code first reads/writes the class
    {
        std::vector<std::vector<int>> fu(static_cast<std::size_t>(S), std::vector<int>(static_cast<std::size_t>(N + 1), N));
        for (int k = N - 1; k >= 0; --k) {
            const GroupPlan::Column& col = pl.cols[static_cast<std::size_t>(k)];
            fu[static_cast<std::size_t>(col.pivSlot)][static_cast<std::size_t>(k)] = k;
            for (int s : col.lSlots) fu[static_cast<std::size_t>(s)][static_cast<std::size_t>(k)] = k;
            for (const GroupPlan::Check& c : col.checks) fu[static_cast<std::size_t>(c.slot)][static_cast<std::size_t>(k)] = k;
            for (const GroupPlan::UEntry& u : col.u) {
                fu[static_cast<std::size_t>(col.pivSlot)][static_cast<std::size_t>(u.j)] = k;
                for (int s : col.lSlots) fu[static_cast<std::size_t>(s)][static_cast<std::size_t>(u.j)] = k;
            }
        }
        for (int r = 0; r < nStage; ++r)
            firstUse[static_cast<std::size_t>(r)] = fu[static_cast<std::size_t>(pl.stageRows[static_cast<std::size_t>(r)].s)][static_cast<std::size_t>(pl.stageRows[static_cast<std::size_t>(r)].j)];
    }
    const int ahead = gopt.stageAhead;
    auto stageReads = [&](int k) {       // reads issued at the head of column k (k = N: before the substitution)
        for (int r = 0; r < nStage; ++r) {
            const int at = ahead < 0 ? 0 : std::max(0, firstUse[static_cast<std::size_t>(r)] - ahead);
            if (at == k && !piped) o << in << "const double sv" << r << " = ST[" << r * G << " + g];\n";
        }
    };
    auto stageAdds = [&](int k) {
        for (int r = 0; r < nStage; ++r) {
            const GroupPlan::StageRow& sr = pl.stageRows[static_cast<std::size_t>(r)];
            const int at = ahead < 0 ? 0 : firstUse[static_cast<std::size_t>(r)];
            if (at == k) o << in << "a_" << sr.s << "_" << sr.j << " += sv" << r << ";\n";
        }
    };

    // ---- elimination
    o << in << "double worst = 0.0, tie = -1.0, pmin = 1.0;   // pivot checks of this lane's rows (see the first column)\n";
    std::vector<std::string> rinv(static_cast<std::size_t>(N));
    for (int k = 0; k < N; ++k) {
        const GroupPlan::Column& col = pl.cols[static_cast<std::size_t>(k)];
        const int sk = col.pivSlot, lk = col.pivLane;
        const std::string ak = "a_" + std::to_string(sk) + "_" + std::to_string(k);
        stageReads(k);
        stageAdds(k);
        o << in << "// column " << k << ": pivot row = lane " << lk << ", slot " << sk << "\n";
        if (col.zeroPivot || col.contradiction) {
            o << in << "worst = 2.0;   // scheduled pivot is a structural zero / contradicts exact constants\n";
            rinv[static_cast<std::size_t>(k)] = "0.0";
            continue;
        }
        std::string absP;
        if (col.pivotConst) {
            absP = lit(std::fabs(col.pivotValue));
            rinv[static_cast<std::size_t>(k)] = lit(1.0 / col.pivotValue);
        } else {
            o << in << "const double pb" << k << " = " << BC << lk << ">(" << ak << ");\n";
            absP = "fabs(pb" + std::to_string(k) + ")";
            o << in << "pmin = fmin(pmin, " << absP << ");\n";          // tiny pivot (solver.hpp:58-61), tested once per solve
        }
        // Candidates (solver.hpp:48-56): every unfinished row with an entry in column k must not be larger than
        // the scheduled pivot, and the rows that come BEFORE it in the reference's scan must be strictly smaller
        // (the first maximum wins a tie).
        //  - not larger: all those rows get a multiplier f = a / pivot anyway, and |a| <= |pivot| is |f| <= 1: one
        //    maximum per slot on the multipliers (below), tested once per solve against 1 + 8 ulp -- the slack
        //    covers the rounding of the refined reciprocal and of the product when |a| == |pivot| exactly (legal
        //    for rows after the pivot).  A row that exceeds the pivot by less than 2e-15 relative is therefore
        //    accepted where the reference would swap: the two pivots then agree to 15 digits, and so do the solves.
        //  - strictly smaller: exact, tie = max(|a| - |pivot|) over exactly those rows must stay < 0.
        // (fmax/fmin drop a NaN operand; a NaN anywhere in the solve makes the update norm non-finite, which is a
        // violation too.)
        std::vector<unsigned> strictOf(static_cast<std::size_t>(S), 0u), anyOf(static_cast<std::size_t>(S), 0u);
        for (const GroupPlan::Check& c : col.checks) {
            anyOf[static_cast<std::size_t>(c.slot)] |= c.laneMask;
            if (c.strict) strictOf[static_cast<std::size_t>(c.slot)] |= c.laneMask;
        }
        for (int s = 0; s < S; ++s) {
            if (!strictOf[static_cast<std::size_t>(s)]) continue;
            char m[16];
            std::snprintf(m, sizeof m, "0x%04x", strictOf[static_cast<std::size_t>(s)] & 0xFFFFu);
            o << in << "tie = fmax(tie, ((" << m << " >> g) & 1) ? fabs(a_" << s << "_" << k << ") - " << absP << " : -1.0);\n";
        }
        if (!col.pivotConst) {
            o << in << "const double r" << k << " = grp_rcp_nr(pb" << k << ");\n";
            rinv[static_cast<std::size_t>(k)] = "r" + std::to_string(k);
        }
        if (col.lSlots.empty()) continue;
        // multipliers (solver.hpp:71); finished rows of the slot being consumed get an exact 0
        for (std::size_t li = 0; li < col.lSlots.size(); ++li) {
            const int s = col.lSlots[li];
            const std::string as = "a_" + std::to_string(s) + "_" + std::to_string(k);
            std::string e;
            const std::string mf = maskName(col.lMask[li]);
            const bool masked = !mf.empty();
            if (col.pivotConst) {
                const double rc = 1.0 / col.pivotValue;
                if (rc == 1.0) e = masked ? as + " * " + mf : as;
                else if (rc == -1.0) e = masked ? "-(" + as + " * " + mf + ")" : "-" + as;
                else e = masked ? as + " * (" + lit(rc) + " * " + mf + ")" : as + " * " + lit(rc);
            } else {
                e = masked ? as + " * (r" + std::to_string(k) + " * " + mf + ")" : as + " * r" + std::to_string(k);
            }
            // The NEGATED multiplier: the updates are then a + nf * u, the accumulate form.  v_fmac_f64 is the one FP64
            // arithmetic instruction with a 4-byte encoding; measured (tools/dev/ubench/valu_lat.hip), a loop of them
            // issues in 4.2 cycles each against 5.1 for v_fma_f64 (8 bytes), and this change alone took the kernel from
            // 2.24e9 to 2.35e9.  Same bits as a - f * u.  (The same trick on the back substitution -- negated
            // reciprocals, negated solution -- converted only 19 more and cost 6 moves + 6 sign flips: 2.29e9, not kept.)
            o << in << "const double nf" << k << "_" << s << " = -(" << e << ");\n";
            if (anyOf[static_cast<std::size_t>(s)]) o << in << "worst = fmax(worst, fabs(nf" << k << "_" << s << "));\n";
        }
        for (const GroupPlan::UEntry& u : col.u) {
            std::string ub;
            if (u.isConst) ub = lit(u.c);
            else {
                o << in << "const double u" << k << "_" << u.j << " = " << BC << lk << ">(a_" << sk << "_" << u.j << ");\n";
                ub = "u" + std::to_string(k) + "_" + std::to_string(u.j);
            }
            for (int s : col.lSlots) {
                const std::string t = "a_" + std::to_string(s) + "_" + std::to_string(u.j);
                if (!declared[static_cast<std::size_t>(s)][static_cast<std::size_t>(u.j)]) return false;   // fill into a class the plan did not mark
                const std::string f = "nf" + std::to_string(k) + "_" + std::to_string(s);
                if (u.isConst && u.c == 1.0) o << in << t << " = " << t << " + " << f << ";\n";
                else if (u.isConst && u.c == -1.0) o << in << t << " = " << t << " - " << f << ";\n";
                else o << in << t << " = " << t << " + " << f << " * " << ub << ";\n";
            }
        }
    }

    stageReads(N);
    stageAdds(N);
    // ---- back substitution, column-wise (solver.hpp:116-128)
    o << in << "// back substitution: x_j is formed in the lane of column j's pivot row, broadcast, and subtracted from the rows\n"
      << in << "// pivoted before it; lane j % 16 keeps it as its solution entry\n";
    for (int s = 0; s < S; ++s) o << in << "double " << xname << s << " = 0.0;\n";
    for (int j = N - 1; j >= 0; --j) {
        const int sj = pl.cols[static_cast<std::size_t>(j)].pivSlot, lj = pl.cols[static_cast<std::size_t>(j)].pivLane;
        const bool home = sj == j / G && lj == j % G;              // the pivot row sits where x_j is kept (first schedule: always)
        o << in << "const double xt" << j << " = a_" << sj << "_" << N << " * " << rinv[static_cast<std::size_t>(j)] << ";\n";
        if (home)
            o << in << xname << sj << " = (g == " << lj << ") ? xt" << j << " : " << xname << sj << ";      // lane " << lj << " keeps its own solution entry\n";
        o << in << "const double xb" << j << " = " << BC << lj << ">(xt" << j << ");\n";
        if (!home)
            o << in << xname << j / G << " = (g == " << j % G << ") ? xb" << j << " : " << xname << j / G << ";\n";
        for (int s : pl.backSlots[static_cast<std::size_t>(j)])
            o << in << "a_" << s << "_" << N << " = a_" << s << "_" << N << " - a_" << s << "_" << j << " * xb" << j << ";\n";
    }
    return true;
    };

    if (!multi) {
        if (!emitSolve(plans[0], in, "xr")) return std::string();
    } else {
        // every schedule gets its own body; a later one runs only while some group's checks failed in all
        // earlier ones (the lane-per-instance kernel does the same, codegen.cpp)
        for (int s = 0; s < S; ++s) o << in << "double xr" << s << " = 0.0;\n";
        o << in << "bool pv = true;\n";
        for (std::size_t a = 0; a < plans.size(); ++a) {
            o << in << (a ? "if (__any(active && pv)) " : "") << "{   // schedule " << a << "\n";
            if (!emitSolve(plans[a], in + "    ", "xa")) return std::string();
            o << in << "    const bool pva = (__ballot(GRP_PIVOTS_BAD) & rowBits) != 0ull;\n";
            for (int s = 0; s < S; ++s) o << in << "    xr" << s << " = pv ? xa" << s << " : xr" << s << ";\n";
            o << in << "    pv = pv && pva;\n"
              << in << "}\n";
        }
    }

    // ---- damped update, norm, convergence (tanalisis.cpp:360-376)
    // The candidate iterate goes to the LDS before it is known whether the solve stands (that needs the norm and
    // the pivot checks): a group whose solve does not stand is a violation and restarts the step from XP elsewhere,
    // and a group that is not iterating keeps its state.
    for (int s = 0; s < S; ++s)
        o << in << "const double xn" << s << " = xo" << s << " + " << lit(K.tran_alpha) << " * (xr" << s << " - xo" << s << ");\n"
          << in << "XS[" << s * G << " + g] = active ? xn" << s << " : xo" << s << ";\n";
    for (int r = 0; r < mosRounds; ++r)
        o << in << "vd" << r << " = XS[mD" << r << "]; vg" << r << " = XS[mG" << r << "]; vs" << r << " = XS[mS" << r << "];\n";
    o << in << "__builtin_amdgcn_sched_barrier(0);      // the reads are issued here, not where the compiler would like them\n";
    o << in << "double ss = 0.0;\n";
    for (int s = 0; s < S; ++s)
        o << in << "{ const double d = xn" << s << " - xo" << s << "; ss += d * d; }\n";
    o << in << (quad ? "ss = grp_sum4(ss);\n" : "ss = grp_sum16(ss);\n")
      // With the guard on, the pass decides on the SQUARED norm: sqrt is monotonic, so `ss < tol^2` and the reference's
      // `sqrt(ss) < tol` (tanalisis.cpp:366-369) can differ only when ss is within a few ulp of tol^2 -- far inside the
      // guard band, where the faithful kernel (which takes the root) has the last word.  Saves the 22-instruction
      // v_rsq_f64 sequence on the critical path of every pass.
      << (guard ? std::string() : in + "const double err = sqrt(ss);\n");
    if (piped) {
        o << in << "__builtin_amdgcn_sched_barrier(0);      // the norm above ran under the reads of the MOSFET inputs\n";
        emitMos(in);
        for (int r = 0; r < nStage; ++r) o << in << "sv" << r << " = ST[" << r * G << " + g];\n";
        o << in << "__builtin_amdgcn_sched_barrier(0);      // the bookkeeping below runs under the reads of the staging rows\n";
    }
    o << (multi ? std::string() : in + "const bool pv = (__ballot(GRP_PIVOTS_BAD) & rowBits) != 0ull;\n")
      << in << "// branch-free bookkeeping (everything here is uniform within a group of 16 lanes)\n"
      << in << "const bool good = active && !pv && (ss < 1.0e300);      // the solve stands: take the damped update\n"
      << in << (guard ? "const bool conv = ss < " + lit(K.tran_tol * K.tran_tol) : "const bool conv = err < " + lit(K.tran_tol)) << ";\n"
      << in << "const bool slow = !conv && iter >= " << (slowIters - 1) << ";                 // slow step: plan.hpp slowStepIters\n";
    if (guard)
        // near-threshold guard: how close did `err < tol` (tanalisis.cpp:369) come to a tie in this step?  Two
        // instructions per pass; everything else happens once per step, below.  (Passes after a group has converged
        // are included: they can only raise a false alarm, which costs a verification and changes nothing.)
        o << in << (gopt.nearForm == 0
                    ? "nearMin = fmin(nearMin, fabs(ss - " + lit(K.tran_tol * K.tran_tol) + "));\n"
                    : "nearAny = nearAny || (ss > " + lit(K.tran_tol * K.tran_tol * (1.0 - 2.0 * gopt.nearBand)) + " && ss < " + lit(K.tran_tol * K.tran_tol * (1.0 + 2.0 * gopt.nearBand)) + ");\n");
    o << in << "viol = viol || (active && !good) || (good && slow);\n"
      << in << "it += good ? 1 : 0;\n";
    for (int s = 0; s < S; ++s) o << in << "xo" << s << " = good ? xn" << s << " : xo" << s << ";\n";
    // the empty asm pins the loads of the next iteration's MOSFET inputs (and staging rows) to this side of the loop's
    // back edge (the compiler otherwise sinks them to the head of the next iteration, in front of what needs them)
    o << in << "active = good && !conv && !slow;\n"
      << in << "__builtin_amdgcn_sched_barrier(0);\n";
    if (piped) {
        for (int r0 = 0; r0 < nStage; r0 += 12) {
            o << in << "asm volatile(\"\" :";
            for (int r = r0; r < std::min(nStage, r0 + 12); ++r) o << (r > r0 ? ", " : " ") << "\"+v\"(sv" << r << ")";
            o << ");\n";
        }
    } else {
        for (int r = 0; r < mosRounds; ++r)
            o << in << "asm volatile(\"\" : \"+v\"(vd" << r << "), \"+v\"(vg" << r << "), \"+v\"(vs" << r << "));\n";
    }
    o
      << "        }\n"      // NR loop
      ;
    if (guard)
        // `err < tol` decided within the rounding noise of this kernel's arithmetic: the group goes on speculatively, the
        // step's start state is kept, and the engine has the faithful kernel verify the step's pass count afterwards.
        // One checkpoint per launch: a second such step stops the group at the start of that step.
        o << "        const bool nearEvent = live && !viol && " << (gopt.nearForm == 0 ? "nearMin <= " + lit(2.0 * gopt.nearBand * K.tran_tol * K.tran_tol) : std::string("nearAny")) << ";\n"
          << "        viol = viol || (nearEvent && nearS != 0);\n";
    o << "        if (live && !viol) {\n";
    if (guard) {
        o << "            if (nearEvent) {           // keep the state at the start of this step for the verification\n"
          << "                nearS = (int)s;\n";
        for (int s = 0; s < S; ++s)
            o << "                if (" << s * G << " + g < " << N << ") nearX[(long long)(" << s * G << " + g) * SB + b] = XP[" << s * G << " + g];\n";
        o << "                if (g == 0) { nearStep[b] = (int)s; nearIt[b] = it; nearItAfter[b] = itTotal; }\n"
          << "            }\n";
    }
    o << "            itTotal += it;\n"
      << "            if (stepIters && g == 0) stepIters[(s - 1) * SB + b] = it;\n"
      << "            if (wave && ophase == 0) {\n"
      << "                const long long row = orow;\n"
      << "                for (int pq = g; pq < nProbe; pq += " << GS << ") wave[(row * nProbe + pq) * SB + b] = XS[probeEq[pq]];\n"
      << "            }\n"
      << "            sdone = s;\n"
      << "        }\n"
      << "    }\n\n"
      << "    if (inb) {\n"
      << "        // a violated instance hands the state at the START of the failing step to the general kernel\n";
    for (int s = 0; s < S; ++s)
        o << "        if (" << s * G << " + g < " << N << ") xio[(long long)(" << s * G << " + g) * SB + b] = viol ? XP[" << s * G << " + g] : xo" << s << ";\n";
    o << "        if (g == 0) {\n"
      << "            if (viol) fallback[b] = 1;\n"
      << "            iters[b] += itTotal;\n"
      << "            status[b] |= st;\n"
      << "            done[b] = (int)sdone;\n"
      << "            if (sdone < nSteps) violFlag[0] = 1;      // unfinished: the engine goes on with this instance\n"
      << (guard ? "            if (nearS != 0 && sdone >= nearS) { nearItAfter[b] = itTotal - nearItAfter[b]; violFlag[1] = 1; }   // to be verified\n" : "")
      << "        }\n"
      << "    }\n"
      << "}\n\n";
    if (quad) o << "}   // namespace csim_q4\n\n";
    return o.str();
}

} // namespace csim
