// Builds the LDS image (MFMA A-operand fragments + fp32 biases) of a network.
//
// What the reference does at this point: SceneNetwork::fillConstantMemory
// (renderer/volume_interpolation_network.cpp:1236-1417) copies the stored half matrices into a
// __constant__ block and the kernel ctor copies them to shared memory
// (renderer/renderer_volume_tensorcores.cuh:401-564).  Here the matrices are permuted ONCE on the
// host into the exact per-lane order v_mfma_f32_32x32x16_f16 wants, so that the accumulator of
// layer l is directly the B operand of layer l+1 (no LDS round trip for activations).
#include "pack.hpp"
#include "srn_device_enums.hpp"

#include <cmath>
#include <cstring>

#include "half.hpp"

namespace fvsrn {

namespace {

// Row of the layer-0 input tile ("X0", produced by the phase MFMA) -> channel index of the padded
// Fourier vector [x,y,z,(t|0),(dx,dy,dz,0),cos(0..F-1),sin(0..F-1)] the reference builds
// (renderer_volume_tensorcores.cuh:768-808).
//  no direction: pass-through channels live in accumulator registers 0,1 of M tile 0, i.e. rows
//                {0,1} (lane half 0) and {4,5} (lane half 1)
//  direction:    registers 0..3 -> rows 0..7 = channels 0..7 (identity)
//  features:     every further even-aligned row pair (2p, 2p+1) = (cos_j, sin_j) of one feature.  Such a pair is a
//                pair of consecutive accumulator registers of ONE lane, which is what the renderer's phase rotation
//                (srn_device.hpp, fourier_advance) needs.
std::vector<int> rowToChannel(int C, bool hasDirection, int numFourier) {
    std::vector<int> map(size_t(C), -1);
    const int base = hasDirection ? 8 : 4;
    if (hasDirection) {
        for (int r = 0; r < 8; ++r) map[size_t(r)] = r;
    } else {
        map[0] = 0; map[1] = 1; map[4] = 2; map[5] = 3;
    }
    int j = 0;
    for (int r = 0; r + 1 < C; r += 2)
        if (map[size_t(r)] < 0 && map[size_t(r + 1)] < 0 && j < numFourier) {
            map[size_t(r)] = base + j;
            map[size_t(r + 1)] = base + numFourier + j;
            ++j;
        }
    if (j != numFourier) throw InvalidNetwork("Fourier feature count does not fill the hidden width");
    return map;
}

struct Frag {  // one A fragment: [lane 0..63][j 0..7] half bits
    uint16_t v[64][8];
};

void putFrag(std::vector<char>& img, size_t off, const Frag& f) { std::memcpy(img.data() + off, f.v, kFragBytes); }

}  // namespace

GridSelection selectGrid(const SceneNetwork& net) {
    // volume_interpolation_network.cpp:1308-1315 (time) and :1332-1334 (ensemble)
    GridSelection g{};
    if (!net.latentGrid) return g;
    const LatentGridTimeAndEnsemble& lg = *net.latentGrid;
    if (lg.hasTimeGrids()) {
        float time = lg.interpolateTime(net.currentTime);
        if (!(time >= 0.f)) time = 0.f;  // NaN-safe (a NaN time would index key frame INT_MIN)
        g.timeIndex = time;
        g.lo = std::min(int(time), lg.timeNum - 1);
        g.hi = std::min(g.lo + 1, lg.timeNum - 1);
        g.frac = time - std::floor(time);
    }
    if (lg.hasEnsembleGrids()) g.ens = lg.interpolateEnsemble(net.currentEnsemble);
    return g;
}

// Key-frame store for the device.  Every key frame is re-laid-out ONCE into x-pair records
//   [Z][Y][X+1][Gc][2] = { v(clamp(xi-1)), v(clamp(xi)) },  xi = floor(texel x) + 1 in [0, X]
// (clamp addressing baked in, so one v_dot2_f32_f16 per channel does the x-lerp of a (z,y) row) and stays RAW
// (fp32 values or bytes): decoding and the A/B time blend happen on the device (grid_blend_kernel), because the
// reference decodes key frame B with A's coefficients (renderer_volume_tensorcores.cuh:586-587).
void packLatentGrid(const SceneNetwork& net, PackedNetwork& out) {
    GridKeyframes& K = out.keys;
    K = GridKeyframes{};
    out.gridMaxAbs.clear();
    out.gridX = out.gridY = out.gridZ = out.gridC = 0;
    if (!net.latentGrid) return;
    const LatentGridTimeAndEnsemble& lg = *net.latentGrid;
    const fvsrn_grid_encoding enc = lg.commonEncoding();
    const LatentGrid& first = lg.hasTimeGrids() ? *lg.timeGrids[0] : *lg.ensembleGrids[0];
    const int X = first.gridSizeX, Y = first.gridSizeY, Z = first.gridSizeZ;
    for (const auto* list : {&lg.timeGrids, &lg.ensembleGrids})
        for (const auto& g : *list)
            if (g->gridSizeX != X || g->gridSizeY != Y || g->gridSizeZ != Z)
                throw Unsupported("all latent grids must share one resolution");
    const int Gt = lg.timeChannels(), Ge = lg.ensembleChannels(), G = Gt + Ge;
    out.gridX = X; out.gridY = Y; out.gridZ = Z; out.gridC = G;
    K.enc = enc; K.X = X; K.Y = Y; K.Z = Z; K.Gt = Gt; K.Ge = Ge;
    K.timeNum = lg.hasTimeGrids() ? lg.timeNum : 0;
    K.ensNum = lg.hasEnsembleGrids() ? lg.ensembleNum : 0;
    K.records = size_t(Z) * Y * (X + 1);
    const size_t esz = enc == FVSRN_GRID_FLOAT ? 4 : 1;
    out.gridMaxAbs.assign(size_t(G), 0.f);

    auto layout = [&](const std::vector<std::shared_ptr<LatentGrid>>& grids, int Gc, int chanBase, std::vector<char>& data,
                      std::vector<float>& off, std::vector<float>& scale, bool timeQuirk) {
        data.assign(grids.size() * K.records * size_t(Gc) * 2 * esz, 0);
        off.assign(grids.size() * size_t(Gc), 0.f);
        scale.assign(grids.size() * size_t(Gc), 1.f);
        for (size_t k = 0; k < grids.size(); ++k) {
            const LatentGrid& g = *grids[k];
            for (int c = 0; c < Gc; ++c) {
                if (enc != FVSRN_GRID_FLOAT) { off[k * Gc + c] = g.gridOffsetOrMean[size_t(c)]; scale[k * Gc + c] = g.gridScaleOrStd[size_t(c)]; }
            }
            char* base = data.data() + k * K.records * size_t(Gc) * 2 * esz;
            for (int z = 0; z < Z; ++z) for (int y = 0; y < Y; ++y) for (int xi = 0; xi <= X; ++xi) {
                const int xs[2] = {std::min(std::max(xi - 1, 0), X - 1), std::min(std::max(xi, 0), X - 1)};
                for (int c = 0; c < Gc; ++c)
                    for (int p = 0; p < 2; ++p) {
                        const size_t i = (((size_t(z) * Y + y) * (X + 1) + xi) * Gc + c) * 2 + p;
                        const size_t src = g.idx(c / 4, z, y, xs[p], c % 4);
                        float r;
                        if (enc == FVSRN_GRID_FLOAT) {
                            r = reinterpret_cast<const float*>(g.grid.data())[src];
                            reinterpret_cast<float*>(base)[i] = r;
                        } else {
                            const uint8_t b = reinterpret_cast<const uint8_t*>(g.grid.data())[src];
                            reinterpret_cast<uint8_t*>(base)[i] = b;
                            r = b / 255.0f;
                        }
                        // range of the decoded values, for the ReLU scaling: own coefficients, and (time grids) the
                        // previous key frame's coefficients, which is what a blend with this frame as "B" uses
                        float m;
                        if (enc == FVSRN_GRID_BYTE_GAUSSIAN) {
                            // |mean| + std * sqrt2 * erfinv(0.99995) bounds every decoded value (bytes 0 / 255: 4.056)
                            constexpr float kMaxGaussian = 4.06f;
                            m = std::fabs(off[k * Gc + c]) + kMaxGaussian * std::fabs(scale[k * Gc + c]);
                            if (timeQuirk && k > 0) m = std::max(m, std::fabs(off[(k - 1) * Gc + c]) + kMaxGaussian * std::fabs(scale[(k - 1) * Gc + c]));
                        } else {
                            m = std::fabs(off[k * Gc + c] + r * scale[k * Gc + c]);
                            if (timeQuirk && k > 0) m = std::max(m, std::fabs(off[(k - 1) * Gc + c] + r * scale[(k - 1) * Gc + c]));
                        }
                        out.gridMaxAbs[size_t(chanBase + c)] = std::max(out.gridMaxAbs[size_t(chanBase + c)], m);
                    }
            }
        }
    };
    if (K.timeNum) layout(lg.timeGrids, Gt, 0, K.timeData, K.timeOffset, K.timeScale, true);
    if (K.ensNum) layout(lg.ensembleGrids, Ge, Gt, K.ensData, K.ensOffset, K.ensScale, false);
}

constexpr int kNoFold = -1000;  // packLayers: no SnakeAlt fold (namespace scope: g++ takes no local variable as a lambda default argument)
PackedNetwork packNetwork(const SceneNetwork& net) {
    std::string why;
    if (!net.valid(&why)) throw InvalidNetwork(why);
    PackedNetwork P;
    P.cfg = net.config();
    const NetworkConfig& c = P.cfg;
    const int C = c.hiddenChannels;
    // every multiple of 16 up to 128 (volume_interpolation_network.cpp:1177-1181 accepts any multiple of 16; wider networks do not fit the
    // reference's 48 KiB of shared memory with more than one hidden layer either, computeMaxWarps :987-1057)
    if (C < 16 || C > 128 || C % 16 != 0)
        throw Unsupported("hidden width " + std::to_string(C) + " is not in the compiled variant set (16, 32, ..., 128)");
    if (c.gridChannels % 16 != 0) throw InvalidNetwork("latent grid channels must be a multiple of 16");
    const int MT = (C + 31) / 32, KS = C / 16, KG = c.gridChannels / 16, KS0 = KS + KG;
    // C->C layers: hidden[L0] (behind Fourier features, or the latent-grid layer) .. hidden[L0+NL-1]; hidden.back() is the
    // last one.  Without Fourier features hidden[0] is the scalar 3|6 -> C first layer (it takes the place of the phase
    // stage, see below) and NL may be 0.
    const int L0 = c.hasFourier ? 0 : 1;
    const int NL = int(net.hidden.size()) - 1 - L0;
    if (NL != c.numHiddenLayers + (c.gridChannels > 0 ? 1 : 0) || NL < (c.hasFourier ? 1 : 0)) throw InvalidNetwork("unexpected layer count");
    if (!c.hasFourier && c.gridChannels > 0) throw InvalidNetwork("a latent grid needs Fourier features");
    const int Cout = net.outputChannelsIn();
    // curvature modes: 6 outputs, of which the DVR path and evaluate() use density + gradient (outputs 0..3)
    if (Cout > 4 && c.outputMode < FVSRN_OUT_DENSITY_CURVATURE) throw Unsupported("more than 4 network outputs are not in the compiled variant set");
    P.MT = MT; P.KS = KS; P.KS0 = KS0; P.NL = NL;

    NetParams& np = P.params;
    size_t off = 0;
    np.offPhase = int(off);  off += size_t(MT) * kFragBytes;
    np.offLayer0 = int(off); off += size_t(NL > 0 ? 1 : 0) * MT * KS0 * kFragBytes;  // [m][s < KS] like a hidden layer, then the latent steps [g][m]
    np.offLast = int(off);   off += size_t(KS) * kFragBytes;
    np.offHidden = int(off); off += size_t(std::max(NL - 1, 0)) * MT * KS * kFragBytes;
    // the kernels prefetch "the next layer" as MT*KS fragments, also when that is the last one (only KS of them are used)
    off = std::max(off, size_t(np.offLast) + size_t(MT) * KS * kFragBytes);
    np.offBias = int(off);   off += (size_t(NL) * 32 * MT + 32 * MT) * sizeof(float);  // same reason: MT blocks for the last layer
    np.ldsBytes = int(off);
    P.ldsImage.assign(off, 0);
    np.numLayers = NL;
    np.gridK = KG;
    np.outputMode = int(c.outputMode);

    const bool hasDir = c.directionMode > 0;
    const int F = c.numFourier;
    // Networks without a time input have a zero in input channel 3 (the padding of the reference's first layer,
    // renderer_volume_tensorcores.cuh:770-782).  Here that channel carries the constant 1 and the first layer's weight column 3 its
    // bias (stored as fp16 like every bias of the reference; 1 x bias is exact in the MFMA), the fp32 bias block of layer 0 is zero:
    // the register-resident kernel with a latent chunk drops those 16 registers (ResidentNet), every other kernel adds a zero.
    const bool foldBias0 = c.hasFourier && !c.passTime && NL > 0;
    np.bias0Folded = foldBias0 ? 1 : 0;
    np.noFourier = c.hasFourier ? 0 : 1;
    std::vector<int> chanOfRow(size_t(C), 0);
    if (c.hasFourier) chanOfRow = rowToChannel(C, hasDir, F);
    else for (int r = 0; r < C; ++r) chanOfRow[size_t(r)] = r;  // rows of the first layer's output = channels
    const int base = hasDir ? 8 : 4;
    const int fcols = c.directionMode == 2 ? 6 : 3;

    // ---- phase fragments: D = Fm * [x,x,y,y,z,z,1,1 | dx,dx,dy,dy,dz,dz,0,0] in revolutions --------
    // Matrix entries are split hi + lo in fp16 (two K slots per input).  The two constant slots carry, also split, minus the
    // integer nearest to the centre of the row's phase range over the unit box (directions in [-1,1]): a whole number of
    // revolutions changes neither cos nor sin, and centred phases stay inside the +-256 revolution domain of v_cos_f32 / v_sin_f32
    // twice as long.  Cosine rows sit on even, sine rows on odd accumulator registers (rowToChannel): the kernels take v_sin_f32 for
    // the odd ones (phase_cos, srn_device.hpp), so both kinds of row have the same, symmetric range -- a NeRF ladder up to 2^9
    // (256 revolutions to either side of the box centre) needs no v_fract in the unshaded renderer.
    double maxPhase = 0, maxPhaseUncentred = 0, maxPhasePlain = 0;
    for (int m = 0; m < MT; ++m) {
        Frag f{};
        for (int lane = 0; lane < 64; ++lane) {
            const int row = 32 * m + (lane & 31), h = lane >> 5;
            if (row >= C) continue;
            const int ch = chanOfRow[size_t(row)];
            float slots[16] = {0};
            if (!c.hasFourier) {
                // scalar first layer (renderer_volume_tensorcores.cuh:810-823): out = b + W p (+ W' d); stored [cin][cout]
                const Layer& L1 = net.hidden[0];
                if (L1.channelsOut != C || L1.channelsIn != (hasDir ? 6 : 3)) throw InvalidNetwork("first layer shape mismatch");
                for (int cin = 0; cin < L1.channelsIn; ++cin)
                    slots[(cin < 3 ? 0 : 8) + 2 * (cin % 3)] = half_bits_to_float(L1.weights[size_t(cin) * C + row]);
                slots[6] = half_bits_to_float(L1.bias[size_t(row)]);
            } else if (ch < base) {
                if (ch < 3) slots[2 * ch] = 1.f;  // position pass-through
                else if (ch == 3) slots[6] = c.passTime ? half_bits_to_float(float_to_half_bits(
                                       net.latentGrid ? net.latentGrid->interpolateTime(net.currentTime) : 0.f)) : (foldBias0 ? 1.f : 0.f);
                else if (ch < 7) slots[8 + 2 * (ch - 4)] = 1.f;  // direction pass-through
            } else {
                const int idx = ch - base;
                const int feat = idx < F ? idx : idx - F;
                double lo = 0, hi = 0;  // phase range over positions in [0,1]^3, directions in [-1,1]^3
                for (int cin = 0; cin < fcols; ++cin) {
                    const double v = double(half_bits_to_float(net.input.fourierMatrix[size_t(feat) + size_t(F) * cin])) /
                                     (2.0 * 3.14159265358979323846);
                    const float vh = half_bits_to_float(float_to_half_bits(float(v)));
                    const float vl = half_bits_to_float(float_to_half_bits(float(v - double(vh))));
                    const int s0 = (cin < 3 ? 0 : 8) + 2 * (cin % 3);
                    slots[s0] = vh;
                    slots[s0 + 1] = vl;
                    if (cin < 3) { lo += std::min(v, 0.0); hi += std::max(v, 0.0); }
                    else { lo -= std::fabs(v); hi += std::fabs(v); }
                }
                const double centre = std::nearbyint(0.5 * (lo + hi));
                const double konst = -centre;
                const float kh = half_bits_to_float(float_to_half_bits(float(konst)));
                slots[6] = kh;
                slots[7] = half_bits_to_float(float_to_half_bits(float(konst - double(kh))));
                maxPhase = std::max(maxPhase, 0.5 * (hi - lo) + 0.75);
                maxPhasePlain = std::max(maxPhasePlain, 0.5 * (hi - lo) + std::fabs(0.5 * (lo + hi) - centre));
                maxPhaseUncentred = std::max(maxPhaseUncentred, std::max(std::fabs(lo), std::fabs(hi)) + 0.25);
            }
            for (int j = 0; j < 8; ++j) f.v[lane][j] = float_to_half_bits(slots[8 * h + j]);
        }
        putFrag(P.ldsImage, size_t(np.offPhase) + size_t(m) * kFragBytes, f);
    }
    if (!c.hasFourier) maxPhase = maxPhaseUncentred = maxPhasePlain = 0;
    // (with a margin of 3/4 revolution: finite-difference gradients sample up to a step outside the box)
    np.fourierNeedsFract = maxPhase >= 255.0 ? 1 : 0;
    // The unshaded renderer evaluates positions inside the box up to the rounding of o + t d (a few 1e-7, i.e. <= 2e-4 revolutions at
    // 2^9): a range of exactly +-256 is taken without v_fract.  Just outside the domain v_cos_f32 returns 1 and v_sin_f32 0 (the ISA's
    // out-of-range results), which at +-(256 + 2e-4) revolutions differ from the true values by 8e-7 and 1.3e-3 -- on the last sample
    // of a ray, in the top octave only.
    np.fourierNeedsFractPlain = maxPhasePlain > 256.0 ? 1 : 0;
    // ... and so that no phase ever lands in that sliver, a network whose range fills the domain has its sample positions clamped to
    // the unit box (three v_med3_f32 per step; the positions leave the box by rounding only, so the clamp moves a sample by <= 1e-6)
    np.fourierClampPos = (!np.fourierNeedsFractPlain && maxPhasePlain > 256.0 - 0.01) ? 1 : 0;
    np.timeSlotOffset = -1;  // set per launch by the device layer (api.cpp, syncTime)
    np.timeSlotBits = 0;
    // evaluate_points takes arbitrary positions: stay exact up to 4 box sizes away without the v_fract
    np.fourierNeedsFractEval = (4.0 * maxPhaseUncentred + std::fabs(maxPhase)) >= 255.0 ? 1 : 0;
    // byte offset of the fp16 "time" entry inside the phase fragment (patched on the device when the time changes):
    // channel 3 sits on row 5 (no direction) / row 3 (direction) of M tile 0, lane half 0, K slot 6
    P.timeSlotOffset = c.passTime ? np.offPhase + (hasDir ? 3 : 5) * 16 + 6 * 2 : -1;

    // The latent grid is packed first: the ReLU scaling below needs the range of its channels.
    packLatentGrid(net, P);
    np.gridX = P.gridX; np.gridY = P.gridY; np.gridZ = P.gridZ; np.gridC = P.gridC;
    np.gridXf = float(P.gridX); np.gridYf = float(P.gridY); np.gridZf = float(P.gridZ);
    // grid_tap (srn_device.hpp) computes record indices in fp32 and byte offsets in 32 bits
    if (double(P.gridZ) * P.gridY * (P.gridX + 1) >= 16777216.0 || double(P.gridZ) * P.gridY * (P.gridX + 1) * P.gridC * 4.0 >= 4294967296.0)
        throw Unsupported("latent grids with 2^24 or more records are not in the compiled variant set");

    // ---- C->C layers + last layer, optionally with power-of-two activation scaling (see reluExponents) -----
    // exps == nullptr: plain image.  exps[l] = e_l: layer l produces h_l * 2^-e_l, i.e. W'_l = W_l * 2^(e_{l-1} - e_l),
    // b'_l = b_l * 2^-e_l, and the last layer multiplies by 2^(e_{NL-1}).  Powers of two: exact in fp16 / fp32.
    auto scaleHalf = [](uint16_t bits, int k) {
        return k == 0 ? bits : float_to_half_bits(std::ldexp(half_bits_to_float(bits), k));
    };
    // foldExp != INT_MIN (SnakeAlt with b = 1/(2p) = 2^foldExp, see ACT_SNAKEALT0): every layer behind an activation takes
    // W' = b W and b' = bias + b * sum_j W_j (fp32), the activation itself leaves out its affine part
    // outShift: the last layer's C-operand rows 4g + o carry output o + outShift (4: the two curvature outputs, see ldsImageCurvature)
    // returns whether the folded first-layer bias (foldBias0) is EXACT in this image: the bias is an fp16 value, but 2^kW x bias may leave the
    // normal range of fp16 in the [0,1]-scaled ReLU image; what the fp16 weight column cannot hold stays in the fp32 bias block of
    // layer 0 (zero as a rule), and an image with such a residue is not run by the register-resident latent-chunk kernel, which drops that block
    auto packLayers = [&](std::vector<char>& img, const std::vector<int>* exps, int foldExp = kNoFold, int outShift = 0) -> bool {
        bool bias0Exact = true;
        float* bias = reinterpret_cast<float*>(img.data() + np.offBias);
        for (int l = 0; l < NL; ++l) {
            const Layer& L = net.hidden[size_t(L0 + l)];
            const int ks = l == 0 ? KS0 : KS;
            const size_t baseOff = l == 0 ? size_t(np.offLayer0) : size_t(np.offHidden) + size_t(l - 1) * MT * KS * kFragBytes;
            if (L.channelsOut != C) throw InvalidNetwork("hidden layer width mismatch");
            const bool fold = foldExp != kNoFold && l > 0;
            const int kW = exps ? (l > 0 ? (*exps)[size_t(l - 1)] : 0) - (*exps)[size_t(l)] : (fold ? foldExp : 0);
            const int kB = exps ? -(*exps)[size_t(l)] : 0;
            for (int m = 0; m < MT; ++m)
                for (int s = 0; s < ks; ++s) {
                    Frag f{};
                    for (int lane = 0; lane < 64; ++lane) {
                        const int row = 32 * m + (lane & 31), h = lane >> 5;
                        if (row >= C) continue;
                        for (int j = 0; j < 8; ++j) {
                            int col;
                            if (s < KS) {
                                const int prevRow = chiOfSlot(16 * s + 8 * h + j);
                                col = l == 0 ? chanOfRow[size_t(prevRow)] : prevRow;
                            } else {
                                col = C + 16 * (s - KS) + 8 * h + j;  // latent grid channel
                            }
                            f.v[lane][j] = scaleHalf(L.weights[size_t(row) * L.channelsIn + col], kW);
                            if (l == 0 && foldBias0 && s < KS && col == 3) f.v[lane][j] = scaleHalf(L.bias[size_t(row)], kW);  // (kW == kB for l = 0)
                        }
                    }
                    const size_t fragIndex = s < KS ? size_t(m) * KS + s : size_t(MT) * KS + size_t(s - KS) * MT + m;
                    putFrag(img, baseOff + fragIndex * kFragBytes, f);
                }
            for (int r = 0; r < C; ++r) {
                double b = std::ldexp(double(half_bits_to_float(L.bias[size_t(r)])), kB);
                if (fold) {
                    double sum = 0;
                    for (int j = 0; j < L.channelsIn; ++j) sum += double(half_bits_to_float(L.weights[size_t(r) * L.channelsIn + j]));
                    b += std::ldexp(sum, foldExp);
                }
                if (l == 0 && foldBias0) {  // b = 2^kB x bias exactly; the weight column holds its fp16 rounding (kW == kB for l = 0)
                    b -= double(half_bits_to_float(scaleHalf(L.bias[size_t(r)], kW)));
                    if (b != 0.0) bias0Exact = false;
                }
                bias[size_t(l) * 32 * MT + r] = float(b);
            }
        }
        // Last layer (C -> 1|4): one v_mfma_f32_16x16x32_f16 per K step and tile -- 16 output rows instead of 32 halve
        // the padding of a 1..4-row matrix.  That instruction reads the SAME B registers differently: lane l supplies
        // column n = l % 16 and K slots 8*(l / 16) .. +7, so column n sees the two samples n (lane groups 0, 2: channel
        // slots 0-7 / 8-15 of the K step) and n + 16 (lane groups 1, 3).  Rows 4g + o of the fragment carry output o for
        // the lanes 16g .. 16g+15 that will read it (D rows 4*(l/16) + r live in lane l): rows of even g take the
        // weights on the K slots of sample n, rows of odd g on those of sample n + 16.  A fragment: lane l = row l % 16,
        // K slots 8*(l / 16) .. +7.
        const Layer& L = net.hidden.back();
        if (L.channelsIn != C || L.channelsOut != Cout) throw InvalidNetwork("last layer shape mismatch");
        const bool transposed = L.channelsIn < 16 || L.channelsOut < 16;  // addLayer stores [in][out]
        const bool foldLast = foldExp != kNoFold && NL > 0;
        const int kL = exps && NL > 0 ? (*exps)[size_t(NL - 1)] : (foldLast ? foldExp : 0);
        for (int s = 0; s < KS; ++s) {
            Frag f{};
            for (int lane = 0; lane < 64; ++lane) {
                const int m = lane & 15, kg = lane >> 4;  // row, K group
                const int g = m >> 2;                     // consumer lane group
                const int o = (m & 3) + outShift;         // output (the renderer and evaluate() take 0..3; curvature: ldsImageCurvature)
                if (o >= Cout) continue;
                if ((kg & 1) != (g & 1)) continue;        // K groups 0,2 belong to sample n, 1,3 to sample n + 16
                const int h = kg >> 1;                    // lane half of the data: channel slots 8h .. 8h+7
                for (int j = 0; j < 8; ++j) {
                    const int col = chiOfSlot(16 * s + 8 * h + j);
                    f.v[lane][j] = scaleHalf(transposed ? L.weights[size_t(col) * Cout + o] : L.weights[size_t(o) * C + col], kL);
                }
            }
            putFrag(img, size_t(np.offLast) + size_t(s) * kFragBytes, f);
        }
        float* bl = bias + size_t(NL) * 32 * MT;
        for (int r = 0; r < 4; ++r) bl[r] = 0.f;
        for (int r = 0; r < 4 && r + outShift < Cout; ++r) {  // C operand rows 4g + r, any g
            const int o = r + outShift;
            double b = double(half_bits_to_float(L.bias[size_t(o)]));
            if (foldLast) {
                double sum = 0;
                for (int col = 0; col < C; ++col) sum += double(half_bits_to_float(transposed ? L.weights[size_t(col) * Cout + o] : L.weights[size_t(o) * C + col]));
                b += std::ldexp(sum, foldExp);
            }
            bl[r] = float(b);
        }
        return bias0Exact;
    };
    packLayers(P.ldsImage, nullptr);  // (plain image: kW = 0, the folded bias is exact)
    // densitycurvature networks: a second image whose last layer computes outputs 4 and 5 (the two curvature values) in rows 0 and 1;
    // only IVolumeInterpolation::evaluateWithGradientAndCurvature reads them (fvsrn_evaluate_points, FVSRN_EVAL_WITH_PREDICTED_CURVATURE)
    P.ldsImageCurvature.clear();
    if (Cout > 4) {
        P.ldsImageCurvature = P.ldsImage;
        packLayers(P.ldsImageCurvature, nullptr, kNoFold, 4);
    }

    // ---- ReLU networks: second image with activations scaled into [0,1] ------------------------------------
    // relu(x) = max(x,0) costs the VALU one v_pk_max_f16 per two values on top of the fp32->fp16 convert.  With
    // h'_l = h_l * 2^-e_l <= 1 guaranteed, "convert + ReLU" is ONE v_cvt_pk_f16_f32 with the clamp modifier
    // (clamp to [0,1]).  e_l comes from interval arithmetic over the actual weights, so the bound is a guarantee,
    // and because all factors are powers of two the scaled network computes bit-for-bit the scaled values (up to
    // fp16 subnormals).  Inputs are bounded for samples inside the box (positions in [0,1], |cos|,|sin| <= 1,
    // latent features by the grid's range), which is what the renderer evaluates; evaluate_points keeps the
    // plain image because callers may pass positions outside the box.
    P.ldsImageScaled.clear();
    P.scaledAct = -1;
    // ---- SnakeAlt with a power-of-two parameter: the affine part of the activation moves into the next layer (ACT_SNAKEALT0) ----
    if (c.activation == FVSRN_ACT_SNAKEALT && c.hasFourier && NL > 0 && c.activationParam > 0) {
        int e2 = 0;
        const double mant = std::frexp(1.0 / (2.0 * double(c.activationParam)), &e2);  // b = mant * 2^e2
        bool ok = mant == 0.5;
        const int foldExp = e2 - 1;
        // exact up to fp16 subnormals (like the scaled ReLU image): a weight below 2^-14 may lose its last bit, an error of at most
        // 2^-25 per weight; a scaled weight must not overflow
        for (int l = 1; l <= NL && ok; ++l)
            for (uint16_t wbits : net.hidden[size_t(L0 + l)].weights) {
                const float w = half_bits_to_float(wbits);
                const float ws = half_bits_to_float(scaleHalf(wbits, foldExp));
                if (!std::isfinite(ws) || std::fabs(double(ws) - std::ldexp(double(w), foldExp)) > std::ldexp(1.0, -25)) { ok = false; break; }
            }
        if (ok) {
            P.ldsImageScaled = P.ldsImage;  // phase fragments, first layer: shared
            P.scaledBias0Exact = packLayers(P.ldsImageScaled, nullptr, foldExp);
            P.scaledAct = ACT_SNAKEALT0;
        }
    }
    if (c.activation == FVSRN_ACT_RELU && c.hasFourier) {
        std::vector<double> bound(size_t(C + c.gridChannels), 1.0005);  // stored input order of layer 0
        if (c.passTime && net.latentGrid) bound[3] = double(std::max(net.latentGrid->timeNum - 1, 0)) + 1e-3;  // any time index
        for (int g = 0; g < c.gridChannels; ++g) bound[size_t(C + g)] = double(P.gridMaxAbs[size_t(g)]) * 1.0005;
        std::vector<int> exps(size_t(NL), 0);
        bool ok = true;
        double maxScaledW = 0;
        for (int l = 0; l < NL && ok; ++l) {
            const Layer& L = net.hidden[size_t(l)];
            std::vector<double> next(size_t(C), 0.0);
            double S = 0;
            for (int r = 0; r < C; ++r) {
                double a = std::fabs(double(half_bits_to_float(L.bias[size_t(r)])));
                for (int j = 0; j < L.channelsIn; ++j)
                    a += std::fabs(double(half_bits_to_float(L.weights[size_t(r) * L.channelsIn + j]))) * bound[size_t(j)];
                next[size_t(r)] = a;
                S = std::max(S, a);
            }
            const int e = std::max(0, int(std::ceil(std::log2(S * 1.0005 + 1e-30))));
            exps[size_t(l)] = e;
            if (e > 12) ok = false;  // beyond this the scaled activations sink into fp16 subnormals
            bound = next;
        }
        if (ok) {
            // scaled weights must stay in the fp16 normal range where it matters
            for (int l = 0; l <= NL; ++l) {
                const Layer& L = net.hidden[size_t(l)];
                const int k = l == NL ? exps[size_t(NL - 1)] : (l > 0 ? exps[size_t(l - 1)] : 0) - exps[size_t(l)];
                for (uint16_t wbits : L.weights) maxScaledW = std::max(maxScaledW, std::fabs(std::ldexp(double(half_bits_to_float(wbits)), k)));
            }
            if (maxScaledW >= 60000.0) ok = false;
        }
        if (ok) {
            P.ldsImageScaled = P.ldsImage;  // phase fragments are shared
            P.scaledBias0Exact = packLayers(P.ldsImageScaled, &exps);
            P.reluExponents = exps;
            P.scaledAct = ACT_RELU01;
        }
    }

    // ---- activation constants (see act() in srn_device.hpp) --------------------------------------------
    const double p = c.activationParam;
    const double pi = 3.14159265358979323846;
    switch (c.activation) {
        case FVSRN_ACT_RELU: np.actA = 0; np.actB = 0; break;
        case FVSRN_ACT_SINE: np.actA = float(p / (2 * pi)); np.actB = 0; break;
        case FVSRN_ACT_SNAKE: np.actA = float(p / pi); np.actB = float(1.0 / (2 * p)); break;
        case FVSRN_ACT_SNAKEALT: np.actA = float(p / pi); np.actB = float(1.0 / (2 * p)); break;
        case FVSRN_ACT_SIGMOID: np.actA = -1.4426950408889634f; np.actB = 0; break;  // exp(-x) = exp2(-log2(e) x)
        default:
            throw Unsupported(std::string("hidden activation ") + activationName(c.activation) +
                              " is not in the compiled variant set (ReLU, Sine, Snake, SnakeAlt, Sigmoid)");
    }
    for (int i = 0; i < 3; ++i) {
        np.boxMin[i] = net.boxMin[i];
        np.boxSize[i] = net.boxSize[i];
        np.invBoxSize[i] = 1.0f / net.boxSize[i];
    }
    P.mfmaFlopsPerSample = mfmaFlopsPerSample(c, int(net.hidden.size()));
    return P;
}

// padded FLOPs per sample that the kernels issue to the matrix cores, from the tiling alone (no packing)
double mfmaFlopsPerSample(const NetworkConfig& c, int numLinearLayers) {
    const int C = c.hiddenChannels;
    // every multiple of 16 up to 128 (volume_interpolation_network.cpp:1177-1181 accepts any multiple of 16; wider networks do not fit the
    // reference's 48 KiB of shared memory with more than one hidden layer either, computeMaxWarps :987-1057)
    if (C < 16 || C > 128 || C % 16 != 0)
        throw Unsupported("hidden width " + std::to_string(C) + " is not in the compiled variant set (16, 32, ..., 128)");
    const int MT = (C + 31) / 32, KS = C / 16, KG = c.gridChannels / 16, KS0 = KS + KG;
    const int NL = numLinearLayers - 1 - (c.hasFourier ? 0 : 1);
    // 32x32x16: 32768 FLOP per 32 samples; the last layer runs 16x16x32 MFMAs (16384 FLOP per 32 samples)
    return 1024.0 * (MT + (NL > 0 ? double(MT) * KS0 : 0.0) + double(std::max(NL - 1, 0)) * MT * KS + 0.5 * KS);
}

}  // namespace fvsrn
