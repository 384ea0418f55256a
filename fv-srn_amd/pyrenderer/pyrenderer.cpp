// pyrenderer -- pybind11 module mirroring the reference's `pyrenderer` surface for the SRN/DVR hot path.
//
// It is a CLIENT of the C ABI (include/fvsrn.h): every class below only holds parameters and forwards to
// libfvsrn.so; torch tensors are used for device memory and the current stream only.
// Reference surface: bindings/bindings.cpp:136-283, renderer/module_registry.cpp:37-106,
// renderer/volume_interpolation_network.cpp:1817-2007, renderer/image_evaluator_simple.cpp:427-475,
// renderer/iimage_evaluator.cpp:325-358, renderer/ray_evaluation_stepping.cpp:80-93,781-801,
// renderer/camera.cpp:184-224,375-397, renderer/volume_interpolation.cpp:615-695, renderer/blending.cpp:31-43,
// renderer/transfer_function*.cpp, renderer/brdf.cpp.
//
// Not mirrored (outside the path): GUI hooks (drawUI), OpenGL rasterisation, Monte-Carlo /
// iso ray evaluators, implicit / grid volumes, importance sampling, compression bindings, interp1D.
#include <torch/extension.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <fstream>
#include <memory>
#include <optional>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/fvsrn.h"

namespace py = pybind11;

namespace {

[[noreturn]] void raise(const std::string& what) { throw std::runtime_error(what); }

void check(int code) {
    if (code != FVSRN_OK) {
        const char* m = fvsrn_last_error();
        raise(m && *m ? std::string(m) : "fvsrn error " + std::to_string(code));
    }
}

void* currentStream() {
    // plumbing: the HIP stream torch is currently recording on (reference: c10::cuda::getCurrentCUDAStream,
    // renderer/iimage_evaluator.cpp:167-170)
    py::object s = py::module_::import("torch").attr("cuda").attr("current_stream")();
    return reinterpret_cast<void*>(s.attr("cuda_stream").cast<uintptr_t>());
}

// ------------------------------------------------------------------------------------------- value types
template <class T> struct V2 { T x{}, y{}; };
template <class T> struct V3 { T x{}, y{}, z{}; };
template <class T> struct V4 { T x{}, y{}, z{}, w{}; };
using float3 = V3<float>; using double3 = V3<double>; using int3 = V3<int>;
using float4 = V4<float>; using double4 = V4<double>; using int2 = V2<int>;

template <class T> void bindV3(py::module_& m, const char* name) {
    using V = V3<T>;
    py::class_<V>(m, name)
        .def(py::init<>())
        .def(py::init([](T x, T y, T z) { return V{x, y, z}; }))
        .def_readwrite("x", &V::x).def_readwrite("y", &V::y).def_readwrite("z", &V::z)
        .def("__str__", [](const V& v) { std::ostringstream s; s << "(" << v.x << ", " << v.y << ", " << v.z << ")"; return s.str(); })
        .def("__add__", [](const V& a, const V& b) { return V{T(a.x + b.x), T(a.y + b.y), T(a.z + b.z)}; })
        .def("__sub__", [](const V& a, const V& b) { return V{T(a.x - b.x), T(a.y - b.y), T(a.z - b.z)}; })
        .def("__mul__", [](const V& a, T s) { return V{T(a.x * s), T(a.y * s), T(a.z * s)}; })
        .def("__rmul__", [](const V& a, T s) { return V{T(a.x * s), T(a.y * s), T(a.z * s)}; });
}
template <class T> void bindV4(py::module_& m, const char* name) {
    using V = V4<T>;
    py::class_<V>(m, name)
        .def(py::init<>())
        .def(py::init([](T x, T y, T z, T w) { return V{x, y, z, w}; }))
        .def_readwrite("x", &V::x).def_readwrite("y", &V::y).def_readwrite("z", &V::z).def_readwrite("w", &V::w)
        .def("__str__", [](const V& v) { std::ostringstream s; s << "(" << v.x << ", " << v.y << ", " << v.z << ", " << v.w << ")"; return s.str(); });
}

// Parameter_<T> (renderer/module_registry.cpp:42-67): scalar value only; tensors / gradients are a training feature
template <class T> struct Parameter {
    T value{};
    bool supportsGradients = false;
};
template <class T> void bindParameter(py::module_& m, const char* name) {
    py::class_<Parameter<T>, std::shared_ptr<Parameter<T>>>(m, name)
        .def_readwrite("value", &Parameter<T>::value)
        .def_readonly("supports_gradients", &Parameter<T>::supportsGradients);
}

// GPUTimer (bindings/bindings.cpp:101-131): an event pair on the current stream
struct GPUTimer {
    py::object e0, e1;
    GPUTimer() {
        py::object ev = py::module_::import("torch").attr("cuda").attr("Event");
        e0 = ev(py::arg("enable_timing") = true);
        e1 = ev(py::arg("enable_timing") = true);
    }
    void start() { e0.attr("record")(); }
    void stop() { e1.attr("record")(); }
    float elapsed() { e1.attr("synchronize")(); return e0.attr("elapsed_time")(e1).cast<float>(); }
};

// --------------------------------------------------------------------------------------------- SceneNetwork
struct SceneNetwork;

struct InputParametrization {
    bool hasTime = false, hasDirection = false;
    int numFourier = 0;
    bool useDirInFourier = false;
    std::vector<float> fourier;  // row-major (F, cols), already premultiplied as given
    bool premultiplied = true;
    std::weak_ptr<SceneNetwork> owner;
    void push();
    int channelsOut() const {
        return numFourier > 0 ? 4 + (hasDirection ? 4 : 0) + 2 * numFourier : 3 + (hasDirection ? 3 : 0);
    }
};

struct OutputParametrization {
    fvsrn_output_mode mode = FVSRN_OUT_DENSITY;
    std::weak_ptr<SceneNetwork> owner;
    void push();
};

struct Layer {  // read-only view of a stored layer
    int channelsIn = 0, channelsOut = 0;
    fvsrn_activation activation = FVSRN_ACT_NONE;
    float activationParameter = 1.f;
};

struct LatentGrid {
    fvsrn_grid_encoding encoding = FVSRN_GRID_FLOAT;
    int gridChannels = 0, gridSizeZ = 0, gridSizeY = 0, gridSizeX = 0;
    std::vector<float> values;  // (C,Z,Y,X) fp32 as given
    bool isValid() const { return gridChannels > 0 && gridChannels % 16 == 0 && gridSizeX > 0 && gridSizeY > 0 && gridSizeZ > 0; }
    static std::shared_ptr<LatentGrid> fromTensor(const torch::Tensor& t_, fvsrn_grid_encoding enc) {
        TORCH_CHECK(t_.dim() == 5 && t_.size(0) == 1, "latent grid tensor must be of shape (1,C,Z,Y,X)");
        torch::Tensor t = t_.detach().to(c10::kCPU, c10::kFloat).contiguous();
        auto g = std::make_shared<LatentGrid>();
        g->encoding = enc;
        g->gridChannels = int(t.size(1)); g->gridSizeZ = int(t.size(2)); g->gridSizeY = int(t.size(3)); g->gridSizeX = int(t.size(4));
        g->values.assign(t.data_ptr<float>(), t.data_ptr<float>() + t.numel());
        return g;
    }
};

struct LatentGridTimeAndEnsemble {
    int timeMin = 0, timeNum = 0, timeStep = 1, ensembleMin = 0, ensembleNum = 0;
    std::vector<std::shared_ptr<LatentGrid>> timeGrids, ensembleGrids;
    fvsrn_network* scratch = nullptr;  // holds the encoded grids so encoding errors can be reported at set time
    LatentGridTimeAndEnsemble() = default;
    LatentGridTimeAndEnsemble(int tmin, int tnum, int tstep, int emin, int enumm)
        : timeMin(tmin), timeNum(tnum), timeStep(tstep), ensembleMin(emin), ensembleNum(enumm),
          timeGrids(size_t(tnum)), ensembleGrids(size_t(enumm)) {
        check(fvsrn_network_create(&scratch));
        check(fvsrn_network_set_latent_grid_layout(scratch, tmin, tnum, tstep, emin, enumm));
    }
    ~LatentGridTimeAndEnsemble() { fvsrn_network_destroy(scratch); }
    LatentGridTimeAndEnsemble(const LatentGridTimeAndEnsemble&) = delete;
    int timeMaxInclusive() const { return timeMin + (timeNum - 1) * timeStep; }
    int ensembleMaxInclusive() const { return ensembleMin + ensembleNum - 1; }
    float interpolateTime(float t) const { return std::min(std::max((t - timeMin) / float(timeStep), 0.f), float(timeNum - 1)); }
    int interpolateEnsemble(int e) const { return std::min(std::max(e - ensembleMin, 0), ensembleNum - 1); }
    double setGrid(bool ensemble, int index, const torch::Tensor& t, fvsrn_grid_encoding enc) {
        auto& list = ensemble ? ensembleGrids : timeGrids;
        TORCH_CHECK(index >= 0 && size_t(index) < list.size(), "index out of bounds!");
        list[size_t(index)] = LatentGrid::fromTensor(t, enc);
        const LatentGrid& g = *list[size_t(index)];
        double err = 0;
        if (!scratch) {
            check(fvsrn_network_create(&scratch));
            check(fvsrn_network_set_latent_grid_layout(scratch, timeMin, timeNum, timeStep, ensembleMin, ensembleNum));
        }
        check(fvsrn_network_set_latent_grid(scratch, ensemble, index, g.values.data(), g.gridChannels, g.gridSizeZ,
                                            g.gridSizeY, g.gridSizeX, enc, &err));
        return err;
    }
    bool isValid() const {
        if (timeGrids.empty() && ensembleGrids.empty()) return false;
        for (auto* l : {&timeGrids, &ensembleGrids})
            for (auto& g : *l)
                if (!g || !g->isValid()) return false;
        return true;
    }
    fvsrn_grid_encoding commonEncoding() const {
        if (!timeGrids.empty() && timeGrids[0]) return timeGrids[0]->encoding;
        if (!ensembleGrids.empty() && ensembleGrids[0]) return ensembleGrids[0]->encoding;
        raise("at least one grid must be active!");
    }
    int timeChannels() const { return timeGrids.empty() || !timeGrids[0] ? 0 : timeGrids[0]->gridChannels; }
    int ensembleChannels() const { return ensembleGrids.empty() || !ensembleGrids[0] ? 0 : ensembleGrids[0]->gridChannels; }
};

struct SceneNetwork : std::enable_shared_from_this<SceneNetwork> {
    fvsrn_network* h = nullptr;
    std::shared_ptr<InputParametrization> input = std::make_shared<InputParametrization>();
    std::shared_ptr<OutputParametrization> output = std::make_shared<OutputParametrization>();
    std::shared_ptr<LatentGridTimeAndEnsemble> latentGrid;
    float3 boxMin{-5.f, -5.f, -5.f}, boxSize{1.f, 1.f, 1.f};  // volume_interpolation_network.cpp:799-800

    SceneNetwork() { check(fvsrn_network_create(&h)); }
    ~SceneNetwork() { fvsrn_network_destroy(h); }
    SceneNetwork(const SceneNetwork&) = delete;
    void link() { input->owner = weak_from_this(); output->owner = weak_from_this(); }

    static std::shared_ptr<SceneNetwork> create() {
        auto n = std::make_shared<SceneNetwork>();
        n->link();
        return n;
    }

    static std::shared_ptr<SceneNetwork> load(const std::string& filename) {
        std::ifstream in(filename, std::ifstream::binary);
        if (!in.is_open()) raise("Unable to open the file " + filename);
        std::vector<char> bytes((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        auto n = std::make_shared<SceneNetwork>();
        fvsrn_network_destroy(n->h);
        n->h = nullptr;
        check(fvsrn_network_create_from_volnet(bytes.data(), bytes.size(), &n->h));
        n->link();
        n->pullFromHandle();
        return n;
    }

    void pullFromHandle() {  // mirror the handle's state into the Python-visible objects
        fvsrn_network_info i;
        check(fvsrn_network_get_info(h, &i));
        input->hasTime = i.has_time; input->hasDirection = i.has_direction; input->numFourier = i.num_fourier;
        input->useDirInFourier = i.use_direction_in_fourier;
        output->mode = fvsrn_output_mode(i.output_mode);
        boxMin = {i.box_min[0], i.box_min[1], i.box_min[2]};
        boxSize = {i.box_size[0], i.box_size[1], i.box_size[2]};
        if (i.grid_channels > 0) {
            latentGrid = std::make_shared<LatentGridTimeAndEnsemble>();
            latentGrid->timeNum = i.time_num; latentGrid->ensembleNum = i.ensemble_num;
        }
        loadedFromFile = true;
    }
    bool loadedFromFile = false;

    void save(const std::string& filename) const {
        size_t n = 0;
        check(fvsrn_network_save_volnet(h, nullptr, 0, &n));
        std::vector<char> buf(n);
        check(fvsrn_network_save_volnet(h, buf.data(), n, &n));
        std::ofstream out(filename, std::ofstream::binary);
        if (!out.is_open()) raise("Unable to open the file " + filename);
        out.write(buf.data(), std::streamsize(n));
    }

    void addLayer(const torch::Tensor& weights, const torch::Tensor& bias, fvsrn_activation act, float param) {
        TORCH_CHECK(weights.dim() == 2, "weights must be (out,in)");
        TORCH_CHECK(bias.dim() == 1 && bias.size(0) == weights.size(0), "bias must be (out)");
        torch::Tensor w = weights.detach().to(c10::kCPU, c10::kFloat).contiguous();
        torch::Tensor b = bias.detach().to(c10::kCPU, c10::kFloat).contiguous();
        check(fvsrn_network_add_layer(h, w.data_ptr<float>(), b.data_ptr<float>(), int(w.size(0)), int(w.size(1)), act, param));
    }
    int numLayers() const {
        fvsrn_network_info i;
        check(fvsrn_network_get_info(h, &i));
        return i.num_layers;
    }
    std::shared_ptr<Layer> getLayer(int index) const {
        auto l = std::make_shared<Layer>();
        int act = 0;
        check(fvsrn_network_get_layer(h, index, &l->channelsOut, &l->channelsIn, &act, &l->activationParameter, nullptr, nullptr));
        l->activation = fvsrn_activation(act);
        return l;
    }
    void setBox(float3 mn, float3 sz) {
        boxMin = mn; boxSize = sz;
        const float a[3] = {mn.x, mn.y, mn.z}, b[3] = {sz.x, sz.y, sz.z};
        check(fvsrn_network_set_box(h, a, b));
    }
    void setLatentGrid(std::shared_ptr<LatentGridTimeAndEnsemble> g) {
        latentGrid = std::move(g);
        if (!latentGrid) return;
        check(fvsrn_network_set_latent_grid_layout(h, latentGrid->timeMin, latentGrid->timeNum, latentGrid->timeStep,
                                                   latentGrid->ensembleMin, latentGrid->ensembleNum));
        for (int e = 0; e < 2; ++e) {
            auto& list = e ? latentGrid->ensembleGrids : latentGrid->timeGrids;
            for (size_t i = 0; i < list.size(); ++i) {
                if (!list[i]) continue;
                const LatentGrid& lg = *list[i];
                check(fvsrn_network_set_latent_grid(h, e, int(i), lg.values.data(), lg.gridChannels, lg.gridSizeZ, lg.gridSizeY,
                                                    lg.gridSizeX, lg.encoding, nullptr));
            }
        }
    }
    bool valid() const {
        const bool ok = fvsrn_network_valid(h) == 1;
        if (!ok) py::print(fvsrn_last_error(), py::arg("file") = py::module_::import("sys").attr("stderr"));
        return ok;
    }
    fvsrn_network_info info() const {
        fvsrn_network_info i;
        check(fvsrn_network_get_info(h, &i));
        return i;
    }
};

void InputParametrization::push() {
    if (auto n = owner.lock())
        check(fvsrn_network_set_input(n->h, hasTime, hasDirection, fourier.empty() ? nullptr : fourier.data(), numFourier,
                                      useDirInFourier ? 6 : 3, premultiplied));
}
void OutputParametrization::push() {
    if (auto n = owner.lock()) check(fvsrn_network_set_output_mode(n->h, mode));
}

// ------------------------------------------------------------------------------------- volume interpolation
struct IVolumeInterpolation {
    virtual ~IVolumeInterpolation() = default;
    double3 boxMin_{-0.5, -0.5, -0.5}, boxMax_{0.5, 0.5, 0.5};
    virtual void setBoxMin(double3 v) { boxMin_ = v; }
    virtual void setBoxMax(double3 v) { boxMax_ = v; }
    double3 boxSize() const { return {boxMax_.x - boxMin_.x, boxMax_.y - boxMin_.y, boxMax_.z - boxMin_.z}; }
    int objectResolution_[3] = {256, 256, 256};  // volume_interpolation.h:39 (networks: the resolution the scene file carries)
    double3 voxelSize() const {  // volume_interpolation.cpp:21-24
        const double3 s = boxSize();
        return {s.x / (objectResolution_[0] - 1), s.y / (objectResolution_[1] - 1), s.z / (objectResolution_[2] - 1)};
    }
    virtual int outputChannels() const = 0;
    // positions in UNIT-BOX coordinates: the reference sets the box to [0,1]^3 for these calls (volume_interpolation.cpp:46-49,154-157)
    virtual torch::Tensor evaluate(const torch::Tensor& positions, const std::optional<torch::Tensor>& direction) = 0;
    // world-space step of the central differences behind evaluate_with_gradients, per axis; {0,0,0}: no gradient available
    virtual double3 gradientStep() const = 0;
    // IVolumeInterpolation::evaluateWithGradient (:128-243, kernel EvaluateNoBatchesWithGradient): densities (N,1) and
    // gradients (N,3) by central differences of evaluate() -- what evalNormal computes for grids (renderer_volume_grid.cuh:234-283:
    // one voxel to either side, scaled by 0.5 / voxel size) and for networks in GRADIENT_MODE_FINITE_DIFFERENCES
    // (renderer_volume_tensorcores.cuh:1185-1196)
    virtual std::tuple<torch::Tensor, torch::Tensor> evaluateWithGradients(const torch::Tensor& positions, const std::optional<torch::Tensor>& direction) {
        if (outputChannels() != 1) raise("evaluateWithGradient can only be called for scalar volumes");
        const double3 h = gradientStep();
        if (!(h.x > 0 && h.y > 0 && h.z > 0)) raise("this volume provides no gradients in its current gradient mode (use FINITE_DIFFERENCES)");
        const torch::Tensor densities = evaluate(positions, direction);
        torch::Tensor gradients = torch::empty({positions.size(0), 3}, densities.options());
        const double hs[3] = {h.x, h.y, h.z};
        for (int k = 0; k < 3; ++k) {
            torch::Tensor offset = torch::zeros({1, 3}, positions.options());
            offset.select(1, k).fill_(hs[k]);
            const torch::Tensor hi = evaluate(positions + offset, direction), lo = evaluate(positions - offset, direction);
            gradients.select(1, k).copy_(((hi - lo) * (0.5 / hs[k])).select(1, 0));
        }
        return {densities, gradients};
    }
    // IVolumeInterpolation::evaluateWithGradientAndCurvature (:245-360): only volumes that estimate the curvature themselves provide it
    // (evalCurvature traps otherwise, renderer_volume_tensorcores.cuh:1541-1556; grids: renderer_volume_grid.cuh has no curvature at all)
    virtual std::tuple<torch::Tensor, torch::Tensor, torch::Tensor> evaluateWithGradientsAndCurvature(const torch::Tensor&,
                                                                                                     const std::optional<torch::Tensor>&) {
        raise("this volume provides no curvature (only networks with output mode densitycurvature / densitycurvature:direct do)");
        return {};
    }
};

enum GradientMode { OFF_OR_DIRECT = 0, FINITE_DIFFERENCES = 1, ADJOINT_METHOD = 2 };
enum Orientation { Xp = 0, Xm = 1, Yp = 2, Ym = 3, Zp = 4, Zm = 5 };

// a volume module this build does not contain ("Grid", "Implicit"): keeps scene files loadable, cannot be rendered
struct UnsupportedVolume : IVolumeInterpolation {
    std::string name;
    explicit UnsupportedVolume(std::string n) : name(std::move(n)) {}
    int outputChannels() const override { return 1; }
    double3 gradientStep() const override { return {0, 0, 0}; }
    torch::Tensor evaluate(const torch::Tensor&, const std::optional<torch::Tensor>&) override {
        raise("volume '" + name + "' is not part of this build (only the SRN path is): set a VolumeInterpolationNetwork");
    }
};

struct VolumeInterpolationNetwork : IVolumeInterpolation {
    std::shared_ptr<SceneNetwork> net;
    bool onlySharedMemory = false;
    GradientMode gradientMode = OFF_OR_DIRECT;
    float finiteDifferencesStepsize = 1 / 256.f;
    float adjointScale = 4;

    void setNetwork(std::shared_ptr<SceneNetwork> n) {  // :1481-1485 + selectNetwork :1448-1472
        if (!n) raise("network is None");
        net = std::move(n);
        const fvsrn_network_info i = net->info();
        boxMin_ = {i.box_min[0], i.box_min[1], i.box_min[2]};
        boxMax_ = {double(i.box_min[0]) + i.box_size[0], double(i.box_min[1]) + i.box_size[1], double(i.box_min[2]) + i.box_size[2]};
    }
    std::shared_ptr<SceneNetwork> currentNetwork() const {
        if (!net) raise("No network loaded");
        return net;
    }
    void pushBox() {  // :1500-1520
        if (!net) return;
        const double3 s = boxSize();
        net->setBox({float(boxMin_.x), float(boxMin_.y), float(boxMin_.z)}, {float(s.x), float(s.y), float(s.z)});
    }
    void setBoxMin(double3 v) override { boxMin_ = v; pushBox(); }
    void setBoxMax(double3 v) override { boxMax_ = v; pushBox(); }
    int outputChannels() const override { return currentNetwork()->info().output_channels; }
    double3 gradientStep() const override {  // unit-box coordinates (see evaluate)
        // (ADJOINT_METHOD does not come here: evaluateWithGradients below takes fvsrn_evaluate_points_adjoint for it)
        if (gradientMode == OFF_OR_DIRECT) return {0, 0, 0};
        return {finiteDifferencesStepsize, finiteDifferencesStepsize, finiteDifferencesStepsize};
    }
    void setTimeAndEnsemble(float t, int e) { check(fvsrn_network_set_time_and_ensemble(currentNetwork()->h, t, e)); }
    // GRADIENT_MODE_OFF_OR_DIRECT on a gradient-predicting network: the network's own gradient outputs (evalNormal,
    // renderer_volume_tensorcores.cuh:1166-1183); otherwise central differences like the base class
    std::tuple<torch::Tensor, torch::Tensor> evaluateWithGradients(const torch::Tensor& positions, const std::optional<torch::Tensor>& direction) override {
        const auto info = currentNetwork()->info();
        const int om = info.output_mode;
        const bool predicts = om >= FVSRN_OUT_DENSITY_GRADIENT && om <= FVSRN_OUT_DENSITY_CURVATURE_DIRECT;
        const bool adjoint = gradientMode == ADJOINT_METHOD;
        if (!adjoint && (gradientMode != OFF_OR_DIRECT || !predicts)) return IVolumeInterpolation::evaluateWithGradients(positions, direction);
        TORCH_CHECK(positions.is_cuda() && positions.dim() == 2 && positions.size(1) == 3, "positions must be a CUDA tensor of shape (N,3)");
        const torch::Tensor p = positions.to(c10::kFloat).contiguous();
        torch::Tensor d;
        if (direction.has_value() && direction->defined()) d = direction->to(c10::kFloat).contiguous();
        torch::Tensor out = torch::empty({p.size(0), 4}, p.options());
        if (adjoint) {
            // GRADIENT_MODE_ADJOINT_METHOD: analytic gradients (evalNormal :1198-1540); latent-grid step as in the renderer
            if (info.output_channels != 1) raise("evaluateWithGradient can only be called for scalar volumes");
            const float gridStep = 1.0f / (float(std::max(info.grid_res[0], 1)) * float(adjointScale));
            check(fvsrn_evaluate_points_adjoint(currentNetwork()->h, p.data_ptr<float>(), d.defined() ? d.data_ptr<float>() : nullptr,
                                                size_t(p.size(0)), out.data_ptr<float>(), gridStep, 0, currentStream()));
        } else
        check(fvsrn_evaluate_points(currentNetwork()->h, p.data_ptr<float>(), d.defined() ? d.data_ptr<float>() : nullptr, size_t(p.size(0)),
                                    out.data_ptr<float>(), FVSRN_EVAL_WITH_PREDICTED_GRADIENT, currentStream()));
        return {out.slice(1, 0, 1).to(positions.scalar_type()).contiguous(), out.slice(1, 1, 4).to(positions.scalar_type()).contiguous()};
    }

    std::tuple<torch::Tensor, torch::Tensor, torch::Tensor> evaluateWithGradientsAndCurvature(const torch::Tensor& positions,
                                                                                             const std::optional<torch::Tensor>& direction) override {
        const int om = currentNetwork()->info().output_mode;
        if (gradientMode != OFF_OR_DIRECT || (om != FVSRN_OUT_DENSITY_CURVATURE && om != FVSRN_OUT_DENSITY_CURVATURE_DIRECT))
            raise("curvature is only supported if the network directly estimates it (output mode densitycurvature*, gradient mode OFF_OR_DIRECT)");
        TORCH_CHECK(positions.is_cuda() && positions.dim() == 2 && positions.size(1) == 3, "positions must be a CUDA tensor of shape (N,3)");
        const torch::Tensor p = positions.to(c10::kFloat).contiguous();
        torch::Tensor d;
        if (direction.has_value() && direction->defined()) d = direction->to(c10::kFloat).contiguous();
        torch::Tensor out = torch::empty({p.size(0), 6}, p.options());
        check(fvsrn_evaluate_points(currentNetwork()->h, p.data_ptr<float>(), d.defined() ? d.data_ptr<float>() : nullptr, size_t(p.size(0)),
                                    out.data_ptr<float>(), FVSRN_EVAL_WITH_PREDICTED_CURVATURE, currentStream()));
        const auto st = positions.scalar_type();
        return {out.slice(1, 0, 1).to(st).contiguous(), out.slice(1, 1, 4).to(st).contiguous(), out.slice(1, 4, 6).to(st).contiguous()};
    }

    torch::Tensor evaluate(const torch::Tensor& positions, const std::optional<torch::Tensor>& direction) override {
        // IVolumeInterpolation::evaluate, renderer/volume_interpolation.cpp:26-127
        TORCH_CHECK(positions.is_cuda(), "positions must reside on the GPU");
        TORCH_CHECK(positions.dim() == 2 && positions.size(1) == 3, "positions must be of shape (N,3)");
        if (direction.has_value() && direction->defined())
            TORCH_CHECK(direction->is_cuda() && direction->dim() == 2 && direction->size(1) == 3, "direction must be a CUDA tensor of shape (N,3)");
        const int channels = outputChannels();
        if (positions.scalar_type() == c10::kHalf) {
            // the scalar-type dispatch of :40-42 for a third type: fp16 positions / directions / values without conversion passes (fvsrn_evaluate_points_half)
            const torch::Tensor ph = positions.contiguous();
            torch::Tensor dh;
            if (direction.has_value() && direction->defined()) dh = direction->to(c10::kHalf).contiguous();
            torch::Tensor outh = torch::empty({ph.size(0), channels}, ph.options());
            check(fvsrn_evaluate_points_half(currentNetwork()->h, ph.data_ptr(), dh.defined() ? dh.data_ptr() : nullptr, size_t(ph.size(0)), outh.data_ptr(), 0,
                                             currentStream()));
            return outh;
        }
        torch::Tensor p = positions.to(c10::kFloat).contiguous();
        torch::Tensor d;
        if (direction.has_value() && direction->defined()) d = direction->to(c10::kFloat).contiguous();
        torch::Tensor out = torch::empty({p.size(0), channels}, p.options());
        check(fvsrn_evaluate_points(currentNetwork()->h, p.data_ptr<float>(), d.defined() ? d.data_ptr<float>() : nullptr,
                                    size_t(p.size(0)), out.data_ptr<float>(), 0, currentStream()));
        return out.to(positions.scalar_type());
    }
};

// --------------------------------------------------------------------------------------------------- grid volumes
// Volume (renderer/volume.h, volume.cpp:1244-1400): host container of named features; .cvol version 1 and the old format, LZ4 or not (save: uncompressed).
struct Volume {
    enum DataType { TypeUChar = 0, TypeUShort = 1, TypeFloat = 2 };
    struct Feature {
        std::string name_;
        DataType type_ = TypeFloat;
        int channels_ = 1;
        int sx = 0, sy = 0, sz = 0;
        std::vector<char> data;  // channel fastest, then x, y, z (MipmapLevel::idx, volume.h:126-132)
        std::string name() const { return name_; }
        DataType type() const { return type_; }
        int channels() const { return channels_; }
        std::tuple<int, int, int> baseResolution() const { return {sx, sy, sz}; }
    };
    float worldX = 1, worldY = 1, worldZ = 1;
    std::vector<std::shared_ptr<Feature>> features;
    static size_t bytesPerType(DataType t) { return t == TypeUChar ? 1 : (t == TypeUShort ? 2 : 4); }

    Volume() = default;
    explicit Volume(const std::string& filename) {  // Volume::Volume(filename), volume.cpp:685-793 + Feature::load :278-332
        // version 1 and the old density-only format, uncompressed or LZ4: the container is read by the C library (fvsrn_cvol_read)
        float world[3] = {1, 1, 1};
        check(fvsrn_cvol_read(filename.c_str(), world, [](void* user, const fvsrn_cvol_feature* info, const void* data, size_t bytes) -> int {
            auto ft = std::make_shared<Feature>();
            ft->name_ = info->name;
            ft->type_ = DataType(info->dtype);
            ft->channels_ = info->channels;
            ft->sx = info->resolution[0]; ft->sy = info->resolution[1]; ft->sz = info->resolution[2];
            ft->data.assign(static_cast<const char*>(data), static_cast<const char*>(data) + bytes);
            static_cast<Volume*>(user)->features.push_back(ft);
            return 0;
        }, this));
        worldX = world[0]; worldY = world[1]; worldZ = world[2];
    }
    void save(const std::string& filename, int compression) const {  // Volume::save, volume.cpp:623-682 (the container is written by the C library)
        std::vector<fvsrn_cvol_feature> infos(features.size());
        std::vector<const void*> ptrs(features.size());
        for (size_t i = 0; i < features.size(); ++i) {
            const Feature& ft = *features[i];
            if (ft.name_.size() >= sizeof(infos[i].name)) raise("feature name too long: " + ft.name_);
            std::memset(&infos[i], 0, sizeof(infos[i]));
            std::memcpy(infos[i].name, ft.name_.data(), ft.name_.size());
            infos[i].index = int(i); infos[i].num_features = int(features.size());
            infos[i].dtype = int(ft.type_); infos[i].channels = ft.channels_;
            infos[i].resolution[0] = ft.sx; infos[i].resolution[1] = ft.sy; infos[i].resolution[2] = ft.sz;
            ptrs[i] = ft.data.data();
        }
        const float world[3] = {worldX, worldY, worldZ};
        check(fvsrn_cvol_write(filename.c_str(), world, int(features.size()), infos.data(), ptrs.data(), compression));
    }
    std::shared_ptr<Feature> getFeature(int index) const {
        if (index < 0 || index >= int(features.size())) raise("feature index out of bounds");
        return features[size_t(index)];
    }
    std::shared_ptr<Feature> getFeatureByName(const std::string& name) const {
        for (const auto& f : features)
            if (f->name_ == name) return f;
        return nullptr;
    }
    // addFeatureFromBuffer (volume.cpp): (C,X,Y,Z) float tensor on the CPU -> float feature
    std::shared_ptr<Feature> addFeatureFromTensor(const std::string& name, const torch::Tensor& t) {
        if (t.scalar_type() != torch::kFloat32) raise("Incompatible format: expected a float array!");
        if (!t.device().is_cpu()) raise("Incompatible format, expected the tensor to reside in CPU memory");
        if (t.dim() != 4) raise("Incompatible buffer dimension, expected a 4D array!");
        auto ft = std::make_shared<Feature>();
        ft->name_ = name;
        ft->type_ = TypeFloat;
        ft->channels_ = int(t.size(0));
        ft->sx = int(t.size(1)); ft->sy = int(t.size(2)); ft->sz = int(t.size(3));
        const torch::Tensor zyxc = t.permute({3, 2, 1, 0}).contiguous();  // z slowest ... channel fastest
        ft->data.resize(size_t(zyxc.numel()) * 4);
        std::memcpy(ft->data.data(), zyxc.data_ptr<float>(), ft->data.size());
        features.push_back(ft);
        return ft;
    }
};

// VolumeInterpolationGrid (renderer/volume_interpolation_grid.cpp): a Volume feature (texture addressing) or a (1,X,Y,Z)
// tensor (accessor addressing) as the density volume of the renderer
struct VolumeInterpolationGrid : IVolumeInterpolation {
    enum VolumeSource { SourceVolume = 0, SourceTorchTensor = 1, SourceEmpty = 2 };
    VolumeSource source_ = SourceEmpty;
    int interpolation_ = FVSRN_VOLUME_TRILINEAR;
    bool newBehavior = false;
    std::shared_ptr<Volume> volume_;
    int mipmapLevel_ = 0;
    torch::Tensor tensor_;
    float minDensity_ = 0, maxDensity_ = 1;
    fvsrn_volume* h = nullptr;
    ~VolumeInterpolationGrid() override { fvsrn_volume_destroy(h); }

    void rebuild(const void* data, int dtype, int sx, int sy, int sz, bool xFastest) {
        fvsrn_volume_destroy(h);
        h = nullptr;
        const double3 s = boxSize();
        const float bmin[3] = {float(boxMin_.x), float(boxMin_.y), float(boxMin_.z)}, bsize[3] = {float(s.x), float(s.y), float(s.z)};
        check(fvsrn_volume_create(data, dtype, sx, sy, sz, xFastest ? 1 : 0, bmin, bsize, &h));
    }
    void setSourceVolume(std::shared_ptr<Volume> v, int mipmap) {  // :136-199
        if (mipmap != 0) raise("mipmap levels are not supported by this build");
        if (!v) { source_ = SourceEmpty; volume_ = nullptr; return; }
        std::shared_ptr<Volume::Feature> density;
        for (const auto& f : v->features)
            if (f->channels_ == 1) { density = f; break; }
        if (!density) raise("Selected volume does not contain any scalar features. Can't render!");
        source_ = SourceVolume;
        volume_ = std::move(v);
        tensor_ = torch::Tensor();
        boxMin_ = {-volume_->worldX / 2.0, -volume_->worldY / 2.0, -volume_->worldZ / 2.0};
        boxMax_ = {volume_->worldX / 2.0, volume_->worldY / 2.0, volume_->worldZ / 2.0};
        rebuild(density->data.data(), int(density->type_), density->sx, density->sy, density->sz, true);
        minDensity_ = 0; maxDensity_ = 1;
    }
    void setSourceTensor(const torch::Tensor& t) {  // :200-224
        TORCH_CHECK(t.dim() == 4, "expected a 4D tensor (B,X,Y,Z)");
        TORCH_CHECK(t.scalar_type() == torch::kFloat32 || t.scalar_type() == torch::kFloat64, "tensor must be of type float or double, but is ", t.dtype());
        TORCH_CHECK(t.size(0) == 1, "batched grids are not supported by this build");
        source_ = SourceTorchTensor;
        volume_ = nullptr;
        mipmapLevel_ = 0;
        tensor_ = t;
        const torch::Tensor host = t.detach().to(torch::kCPU, torch::kFloat32).contiguous();
        minDensity_ = host.min().item<float>();
        maxDensity_ = host.max().item<float>();
        const int X = int(t.size(1)), Y = int(t.size(2)), Z = int(t.size(3));
        const double voxel = 1.0 / std::max({X, Y, Z});
        boxMin_ = {-X * voxel / 2.0, -Y * voxel / 2.0, -Z * voxel / 2.0};
        boxMax_ = {X * voxel / 2.0, Y * voxel / 2.0, Z * voxel / 2.0};
        rebuild(host.data_ptr<float>(), FVSRN_VOLUME_F32, X, Y, Z, false);
    }
    void pushBox() {  // the C ABI volume carries its box: re-create it from the current source
        if (source_ == SourceTorchTensor && tensor_.defined()) {
            const torch::Tensor host = tensor_.detach().to(torch::kCPU, torch::kFloat32).contiguous();
            rebuild(host.data_ptr<float>(), FVSRN_VOLUME_F32, int(host.size(1)), int(host.size(2)), int(host.size(3)), false);
        } else if (source_ == SourceVolume && volume_) {
            for (const auto& f : volume_->features)
                if (f->channels_ == 1) { rebuild(f->data.data(), int(f->type_), f->sx, f->sy, f->sz, true); break; }
        }
    }
    void setBoxMin(double3 v) override { boxMin_ = v; pushBox(); }
    void setBoxMax(double3 v) override { boxMax_ = v; pushBox(); }
    int outputChannels() const override { return 1; }
    fvsrn_volume* handle() const {
        if (source_ == SourceEmpty || !h) raise("No volume specified, can't render!");
        return h;
    }
    int sourceMode() const { return source_ == SourceTorchTensor ? FVSRN_VOLUME_SOURCE_TENSOR : FVSRN_VOLUME_SOURCE_TEXTURE; }
    double3 gradientStep() const override {  // one voxel (normalStep = 1, normalScale = 0.5 / voxelSize, :1097-1104), in unit-box units
        int res[3];
        check(fvsrn_volume_info(handle(), res, nullptr, nullptr));
        const int m = newBehavior ? 0 : 1;
        return {1.0 / (res[0] - m), 1.0 / (res[1] - m), 1.0 / (res[2] - m)};
    }
    torch::Tensor evaluate(const torch::Tensor& positions, const std::optional<torch::Tensor>&) override {
        TORCH_CHECK(positions.is_cuda(), "positions must reside on the GPU");
        TORCH_CHECK(positions.dim() == 2 && positions.size(1) == 3, "positions must be of shape (N,3)");
        // unit-box positions (the reference evaluates with the box set to [0,1]^3) -> the world box of the C ABI volume
        const double3 bs = boxSize();
        const torch::Tensor scale = torch::tensor({float(bs.x), float(bs.y), float(bs.z)}, positions.options().dtype(c10::kFloat));
        const torch::Tensor shift = torch::tensor({float(boxMin_.x), float(boxMin_.y), float(boxMin_.z)}, positions.options().dtype(c10::kFloat));
        const torch::Tensor p = (positions.to(c10::kFloat) * scale + shift).contiguous();
        torch::Tensor out = torch::empty({p.size(0), 1}, p.options());
        check(fvsrn_volume_evaluate_points(handle(), sourceMode(), interpolation_, newBehavior ? 1 : 0, p.data_ptr<float>(), size_t(p.size(0)),
                                           out.data_ptr<float>(), currentStream()));
        return out.to(positions.scalar_type());
    }
};

// ----------------------------------------------------------------------------------------------- TF / BRDF
struct ITransferFunction {
    virtual ~ITransferFunction() = default;
    virtual void fill(fvsrn_scene_desc& d, std::vector<float>& table) const = 0;
    // ITransferFunction::evaluate / evaluate_with_previous (transfer_function.cpp:132-145): densities (B,1) -> colours (B,4)
    torch::Tensor evaluate(const torch::Tensor& densities, double densityMin, double densityMax, const std::optional<torch::Tensor>& previous,
                           double stepsize) const {
        TORCH_CHECK(densities.dim() == 2 && densities.size(1) == 1, "densities must be of shape (B,1)");
        TORCH_CHECK(densities.is_cuda() && densities.scalar_type() == torch::kFloat32, "densities must be a float32 tensor on the GPU");
        fvsrn_scene_desc d{};
        std::vector<float> table;
        fill(d, table);
        d.tf_table = table.empty() ? nullptr : table.data();
        d.cam_right[0] = 1; d.cam_up[1] = 1; d.fov_y_radians = 1; d.stepsize = 1; d.density_max = 1;  // a valid (unused) rest of the scene
        fvsrn_scene* sc = nullptr;
        check(fvsrn_scene_create(&d, &sc));
        const torch::Tensor in = densities.contiguous();
        torch::Tensor prev;
        if (previous) {
            TORCH_CHECK(previous->sizes() == densities.sizes() && previous->is_cuda() && previous->scalar_type() == torch::kFloat32,
                        "previous_density must match densities");
            prev = previous->contiguous();
        }
        torch::Tensor out = torch::empty({in.size(0), 4}, in.options());
        const int rc = fvsrn_scene_evaluate_tf(sc, in.data_ptr<float>(), previous ? prev.data_ptr<float>() : nullptr, size_t(in.size(0)),
                                               float(densityMin), float(densityMax), float(stepsize), out.data_ptr<float>(), currentStream());
        fvsrn_scene_destroy(sc);
        check(rc);
        return out;
    }
};
struct TransferFunctionIdentity : ITransferFunction {
    // transfer_function_identity.cpp:34-38: absorption_emission = (scaleAbsorption, scaleEmission)
    std::shared_ptr<Parameter<V2<double>>> absorptionEmission = std::make_shared<Parameter<V2<double>>>();
    TransferFunctionIdentity() { absorptionEmission->value = {1.0, 1.0}; }
    void fill(fvsrn_scene_desc& d, std::vector<float>&) const override {
        d.tf_kind = FVSRN_TF_IDENTITY;
        d.tf_scale_absorption = float(absorptionEmission->value.x);
        d.tf_scale_emission = float(absorptionEmission->value.y);
    }
};
struct TableTF : ITransferFunction {
    torch::Tensor tensor;  // (1,R,cols) like the reference's textureTensor_
    int kind, cols;
    TableTF(int k, int c) : kind(k), cols(c) {}
    void setTensor(const torch::Tensor& t) {
        TORCH_CHECK(t.dim() == 3 && t.size(0) == 1 && t.size(2) == cols, "TF tensor must be of shape (1,R,", cols, ")");
        tensor = t.detach().to(c10::kCPU, c10::kFloat).contiguous();
    }
    void fill(fvsrn_scene_desc& d, std::vector<float>& table) const override {
        if (!tensor.defined()) raise("the transfer function has no control points");
        d.tf_kind = kind;
        d.tf_rows = int(tensor.size(1));
        table.assign(tensor.data_ptr<float>(), tensor.data_ptr<float>() + tensor.numel());
    }
};
struct TransferFunctionGaussian : TableTF {
    double absorptionScaling = 1.0;            // transfer_function_gaussian.cpp:264 (the tensor rows already carry opacity * scaling)
    bool piecewiseAnalyticIntegration = false;  // :265  usePiecewiseAnalyticIntegration_  (TRANSFER_FUNCTION_GAUSSIAN__ANALYTIC)
    bool scaleWithGradient = false;             // :238  scaleWithGradient_ (JSON only in the reference; a property here as well)
    TransferFunctionGaussian() : TableTF(FVSRN_TF_GAUSSIAN, 6) {}
    void fill(fvsrn_scene_desc& d, std::vector<float>& table) const override {
        if (scaleWithGradient && piecewiseAnalyticIntegration)  // getDefines, :296-297
            raise("Gaussian TF: gradient scaling and piecewise analytic integration are incompatible");
        TableTF::fill(d, table);
        d.tf_gaussian_mode = scaleWithGradient ? FVSRN_TF_GAUSSIAN_SCALE_WITH_GRADIENT
                                               : (piecewiseAnalyticIntegration ? FVSRN_TF_GAUSSIAN_ANALYTIC : FVSRN_TF_GAUSSIAN_PLAIN);
    }
};
struct TransferFunctionPiecewise : TableTF { TransferFunctionPiecewise() : TableTF(FVSRN_TF_PIECEWISE, 5) {} };
struct TransferFunctionTexture : TableTF {
    int preintegrationMode = FVSRN_PREINTEGRATE_NONE;  // TransferFunctionTexture::PreintegrationMode
    TransferFunctionTexture() : TableTF(FVSRN_TF_TEXTURE, 4) {}
    void fill(fvsrn_scene_desc& d, std::vector<float>& table) const override {
        TableTF::fill(d, table);
        d.tf_preintegration = preintegrationMode;
    }
};

struct IBRDF { virtual ~IBRDF() = default; };
struct BRDFLambert : IBRDF {  // renderer/brdf.cpp:208-225 (JSON), :413-508 (constant block, light follows camera)
    bool enableMagnitudeScaling = false, enablePhong = false;
    std::shared_ptr<Parameter<double>> magnitudeScaling = std::make_shared<Parameter<double>>(Parameter<double>{1.0});
    std::shared_ptr<Parameter<double>> ambient = std::make_shared<Parameter<double>>(Parameter<double>{0.1});
    std::shared_ptr<Parameter<double>> specular = std::make_shared<Parameter<double>>(Parameter<double>{0.1});
    std::shared_ptr<Parameter<double>> magnitudeCenter = std::make_shared<Parameter<double>>(Parameter<double>{0.5});
    std::shared_ptr<Parameter<double>> magnitudeRadius = std::make_shared<Parameter<double>>(Parameter<double>{0.1});
    std::shared_ptr<Parameter<int>> specularExponent = std::make_shared<Parameter<int>>(Parameter<int>{16});
    bool lightFollowsCamera = true;
    int lightType = FVSRN_LIGHT_DIRECTIONAL;
    std::shared_ptr<Parameter<double3>> lightPosition = std::make_shared<Parameter<double3>>();
    std::shared_ptr<Parameter<double3>> lightDirection = std::make_shared<Parameter<double3>>();
};

struct Blending {
    int blendMode = FVSRN_BLEND_BEER_LAMBERT;  // blending.h:54 default
};

// ---------------------------------------------------------------------------------------------- ray evaluator
struct IRayEvaluation { virtual ~IRayEvaluation() = default; };
struct IRayEvaluationStepping : IRayEvaluation {  // ray_evaluation_stepping.cpp:80-93
    double stepsize = 0.005;
};
struct RayEvaluationSteppingDvr : IRayEvaluationStepping {
    double minDensity = 0.0, maxDensity = 1.0;
    bool enableEarlyOut = true;
    std::shared_ptr<Blending> blending = std::make_shared<Blending>();
    std::shared_ptr<ITransferFunction> tf = std::make_shared<TransferFunctionIdentity>();
    std::shared_ptr<BRDFLambert> brdf = std::make_shared<BRDFLambert>();
    void convertToTextureTF();  // ray_evaluation_stepping.cpp:767-779
};

// ----------------------------------------------------------------------------------------------------- camera
struct ICamera {
    virtual ~ICamera() = default;
    double aspectRatio = 1.0;
    virtual void frame(float eye[3], float right[3], float up[3], int batch = 0) = 0;
    virtual int batches() const { return 1; }  // ICamera::getBatches (imodule.h): B of externally set (B,3,3) camera matrices
    double fovYRadians = 45.0 * 3.14159265358979323846 / 180.0;
};
struct CameraOnASphere : ICamera {
    int orientation = 3;  // Ym, camera.cpp:225-233
    std::shared_ptr<Parameter<double3>> center = std::make_shared<Parameter<double3>>();
    std::shared_ptr<Parameter<double3>> pitchYawDistance = std::make_shared<Parameter<double3>>();
    torch::Tensor external;  // (B,3,3) from set_parameters
    CameraOnASphere() { pitchYawDistance->value = {0, 0, 1}; }
    int batches() const override { return external.defined() && external.numel() > 0 ? int(external.size(0)) : 1; }
    void frame(float eye[3], float right[3], float up[3], int batch = 0) override {
        if (external.defined() && external.numel() > 0) {
            // (B,3,3) reference frames set from outside (camera.cpp:242-258): batch entry b renders with matrix b -- the batch dimension
            // virtual_size.z of ImageEvaluatorSimpleKernel (renderer_image_evaluator_simple.cuh:36-127) is a host loop over launches here
            TORCH_CHECK(batch >= 0 && batch < external.size(0), "camera batch index out of range");
            torch::Tensor m = external.to(c10::kCPU, c10::kFloat).contiguous();
            const float* p = m.data_ptr<float>() + 9 * batch;
            for (int i = 0; i < 3; ++i) { eye[i] = p[i]; right[i] = p[3 + i]; up[i] = p[6 + i]; }
            return;
        }
        const double c[3] = {center->value.x, center->value.y, center->value.z};
        check(fvsrn_camera_on_a_sphere(orientation, c, pitchYawDistance->value.x, pitchYawDistance->value.y,
                                       pitchYawDistance->value.z, eye, right, up));
    }
    torch::Tensor getParameters() {
        const int B = batches();
        torch::Tensor m = torch::empty({B, 3, 3}, torch::kFloat);
        for (int b = 0; b < B; ++b) {
            float e[3], r[3], u[3];
            frame(e, r, u, b);
            float* p = m.data_ptr<float>() + 9 * b;
            for (int i = 0; i < 3; ++i) { p[i] = e[i]; p[3 + i] = r[i]; p[6 + i] = u[i]; }
        }
        return m.to(torch::kCUDA);
    }
    void setParameters(const torch::Tensor& t) {
        if (!t.defined() || t.numel() == 0) { external = torch::Tensor(); return; }
        TORCH_CHECK(t.dim() == 3 && t.size(1) == 3 && t.size(2) == 3, "camera matrix must be of shape B,3,3, but is of shape", t.sizes());
        external = t;
    }
    double3 getOrigin(int batch) {
        TORCH_CHECK(batch == 0, "getOrigin is only available for batch=0 (for now)");
        float e[3], r[3], u[3];
        frame(e, r, u);
        return {e[0], e[1], e[2]};
    }
    double3 getFront(int batch) {
        TORCH_CHECK(batch == 0, "getFront is only available for batch=0 (for now)");
        float e[3], r[3], u[3];
        frame(e, r, u);
        return {double(u[1] * r[2] - u[2] * r[1]), double(u[2] * r[0] - u[0] * r[2]), double(u[0] * r[1] - u[1] * r[0])};
    }
};

// -------------------------------------------------------------------------------------------- image evaluator
enum ChannelMode { ChannelMask = 0, ChannelNormal = 1, ChannelDepth = 2, ChannelColor = 3 };

struct ImageEvaluatorSimple {
    std::shared_ptr<ICamera> camera = std::make_shared<CameraOnASphere>();
    std::shared_ptr<IRayEvaluation> rayEvaluator = std::make_shared<RayEvaluationSteppingDvr>();
    std::shared_ptr<IVolumeInterpolation> volume;
    ChannelMode selectedChannel = ChannelColor;
    bool doublePrecision = false;
    int sppLog2 = 0;
    bool useTonemapping = false;
    float tonemappingShoulder = 1.0f, lastMaxExposure = 1.0f, fixedMaxExposure = 1.0f;
    bool fixMaxExposure = false;
    fvsrn_scene* scene = nullptr;
    ~ImageEvaluatorSimple() { fvsrn_scene_destroy(scene); }

    torch::Tensor render(int width, int height) { return renderRows(width, height, -1, 1, 16); }  // image_evaluator_simple.cpp:198-361

    // Multi-GPU frames behind the module API (no reference counterpart: its render() owns the whole frame on one device).  Rank `rank`
    // of `world` renders the image rows y with (y / stripe) % world == rank -- round-robin stripes balance empty and dense image regions -- of the SAME
    // scene (camera, TF, network: every rank holds a copy) into a compact (B, 8, rows, W) tensor, rows = ImageEvaluatorSimple.stripe_rows(...).  The
    // caller moves the compact tensors with the collective of its choice (torch.distributed all_gather_into_tensor / gather over RCCL) and puts them
    // back in image order with ImageEvaluatorSimple.Assemble_stripes; fv-srn_amd/tiles.py StripeRenderer is that pipeline ready-made.
    torch::Tensor renderStripes(int width, int height, int rank, int world, int stripe) {
        if (world < 1 || rank < 0 || rank >= world) raise("render_stripes: 0 <= rank < world");
        if (stripe <= 0 || stripe % 8 != 0) raise("render_stripes: stripe must be a positive multiple of 8 (the pixel tile of one wave)");
        if (!std::dynamic_pointer_cast<VolumeInterpolationNetwork>(volume)) raise("render_stripes needs a VolumeInterpolationNetwork (grid volumes render whole frames)");
        return renderRows(width, height, rank, world, stripe);
    }
    static int stripeRows(int height, int stripe, int rank, int world) { return fvsrn_stripe_rows(height, stripe, rank, world); }
    // (world, B, 8, rows, W) compact stripe images of all ranks (the layout all_gather_into_tensor / gather produce) -> (B, 8, H, W)
    static torch::Tensor assembleStripes(const torch::Tensor& gathered, int height, int stripe) {
        TORCH_CHECK(gathered.dim() == 5 && gathered.size(2) == 8, "gathered must be of shape (world, B, 8, rows, W)");
        const int64_t world = gathered.size(0), B = gathered.size(1), rows = gathered.size(3), W = gathered.size(4);
        TORCH_CHECK(stripe > 0 && rows * world == height && height % (int64_t(stripe) * world) == 0,
                    "height must be a multiple of stripe * world (every rank owns the same number of rows)");
        return gathered.contiguous().view({world, B, 8, rows / stripe, stripe, W}).permute({1, 2, 3, 0, 4, 5}).reshape({B, 8, height, W});
    }

    // rank < 0: the whole frame; else this rank's stripes, compact
    torch::Tensor renderRows(int width, int height, int rank, int world, int stripe) {
        auto vol = std::dynamic_pointer_cast<VolumeInterpolationNetwork>(volume);
        auto grid = std::dynamic_pointer_cast<VolumeInterpolationGrid>(volume);
        if (!vol && !grid) {
            auto other = std::dynamic_pointer_cast<UnsupportedVolume>(volume);
            raise(std::string("ImageEvaluatorSimple.volume must be a VolumeInterpolationNetwork or a VolumeInterpolationGrid") +
                  (other ? "; the scene file selected volume '" + other->name + "'" : ""));
        }
        auto dvr = std::dynamic_pointer_cast<RayEvaluationSteppingDvr>(rayEvaluator);
        if (!dvr) raise("ImageEvaluatorSimple.ray_evaluator must be a RayEvaluationSteppingDvr");
        if (!camera) raise("no camera selected");
        if (doublePrecision) raise("double precision rendering is not supported by the SRN path");
        camera->aspectRatio = double(width) / height;
        // computeBatchCount (iimage_evaluator.cpp:138-165): only the camera can carry a batch dimension here (TF / step-size tensors: B = 1)
        const int B = camera->batches();
        const int rows = rank < 0 ? height : fvsrn_stripe_rows(height, stripe, rank, world);
        torch::Tensor out = torch::empty({B, 8, rows, width}, torch::TensorOptions().dtype(torch::kFloat).device(torch::kCUDA));
        for (int batch = 0; batch < B; ++batch) {
        fvsrn_scene_desc d{};
        std::vector<float> table;
        camera->frame(d.cam_eye, d.cam_right, d.cam_up, batch);
        d.fov_y_radians = float(camera->fovYRadians);
        d.stepsize = float(dvr->stepsize);
        d.density_min = float(dvr->minDensity);
        d.density_max = float(dvr->maxDensity);
        d.early_out = dvr->enableEarlyOut;
        d.blend_mode = dvr->blending ? dvr->blending->blendMode : FVSRN_BLEND_BEER_LAMBERT;
        bool rgbo = false;
        if (vol) {
            const fvsrn_network_info info = vol->currentNetwork()->info();
            rgbo = info.output_mode == FVSRN_OUT_RGBO || info.output_mode == FVSRN_OUT_RGBO_DIRECT;
        }
        if (rgbo) {
            d.tf_kind = FVSRN_TF_NONE;  // ray_evaluation_stepping.cpp:560-601: colour volumes skip the TF
        } else {
            if (!dvr->tf) raise("no transfer function selected");
            dvr->tf->fill(d, table);
        }
        d.tf_table = table.empty() ? nullptr : table.data();
        d.gradient_mode = !vol ? FVSRN_GRADIENT_OFF_OR_DIRECT
                               : (vol->gradientMode == FINITE_DIFFERENCES ? FVSRN_GRADIENT_FINITE_DIFFERENCES
                                  : (vol->gradientMode == ADJOINT_METHOD ? FVSRN_GRADIENT_ADJOINT_METHOD : FVSRN_GRADIENT_OFF_OR_DIRECT));
        d.finite_differences_stepsize = vol ? float(vol->finiteDifferencesStepsize) : 0.f;
        if (vol && vol->gradientMode == ADJOINT_METHOD) {  // VolumeInterpolationNetwork::fillConstantMemory :1808-1812
            const fvsrn_network_info info = vol->currentNetwork()->info();
            d.adjoint_grid_stepsize = 1.0f / (float(std::max(info.grid_res[0], 1)) * float(vol->adjointScale));
        }
        if (dvr->brdf) {  // BRDFLambert::fillConstantMemory, brdf.cpp:413-448
            BRDFLambert& b = *dvr->brdf;
            if (b.lightFollowsCamera) {  // updateLightFromCamera :490-508: camera origin / front = cross(up, right)
                const float* e = d.cam_eye; const float* r = d.cam_right; const float* u = d.cam_up;
                b.lightPosition->value = double3{e[0], e[1], e[2]};
                b.lightDirection->value = double3{double(u[1]) * r[2] - double(u[2]) * r[1], double(u[2]) * r[0] - double(u[0]) * r[2],
                                                  double(u[0]) * r[1] - double(u[1]) * r[0]};
            }
            d.brdf_enable_magnitude_scaling = b.enableMagnitudeScaling;
            d.brdf_enable_phong = b.enablePhong;
            d.brdf_magnitude_scaling = float(b.magnitudeScaling->value);
            d.brdf_ambient = float(b.ambient->value);
            d.brdf_specular = float(b.specular->value);
            d.brdf_magnitude_center = float(b.magnitudeCenter->value);
            d.brdf_magnitude_radius = float(b.magnitudeRadius->value);
            d.brdf_specular_exponent = b.specularExponent->value;
            d.brdf_light_type = b.lightType;
            const double3 l = b.lightType == FVSRN_LIGHT_POINT ? b.lightPosition->value : b.lightDirection->value;
            d.brdf_light[0] = float(l.x); d.brdf_light[1] = float(l.y); d.brdf_light[2] = float(l.z);
        }
        if (!scene) check(fvsrn_scene_create(&d, &scene));
        else check(fvsrn_scene_update(scene, &d));
        float* dst = out.data_ptr<float>() + size_t(batch) * 8 * size_t(rows) * size_t(width);
        // The batch dimension B of ImageEvaluatorSimpleKernel (virtual_size.z, renderer_image_evaluator_simple.cuh:36-127: ONE launch over all batch
        // entries) as one call into the library: fvsrn_render_stripes_batch renders up to eight camera poses per launch (a work unit is (frame, pixel
        // tile): r05).  Scenes whose light follows the camera have a per-entry scene description and keep one launch per entry.
        if (vol && B > 1 && batch == 0 && !(dvr->brdf && dvr->brdf->lightFollowsCamera)) {
            std::vector<float> cams(size_t(B) * 9);
            for (int b = 0; b < B; ++b) camera->frame(&cams[size_t(b) * 9], &cams[size_t(b) * 9 + 3], &cams[size_t(b) * 9 + 6], b);
            void* st = currentStream();
            check(fvsrn_render_stripes_batch(&scene, &st, 1, vol->currentNetwork()->h, width, height, rank >= 0 ? stripe : 8, rank >= 0 ? rank : 0,
                                             rank >= 0 ? world : 1, B, cams.data(), nullptr, out.data_ptr<float>(), nullptr, 0, 1.0f, nullptr));
            break;
        }
        if (vol && rank >= 0)
            check(fvsrn_render_stripes(scene, vol->currentNetwork()->h, width, height, stripe, rank, world, dst, nullptr, currentStream()));
        else if (vol)
            check(fvsrn_render(scene, vol->currentNetwork()->h, width, height, 0, height, dst, nullptr, currentStream()));
        else
            check(fvsrn_render_volume(scene, grid->handle(), grid->sourceMode(), grid->interpolation_, grid->newBehavior ? 1 : 0,
                                      selectedChannel == ChannelNormal ? 1 : 0 /* image_evaluator_simple.cpp:249-252 */, width, height,
                                      dst, nullptr, currentStream()));
        }  // batch
        if (rank >= 0) return out;  // (a rank's stripes: exposure / refinement state belongs to whole frames)
        lastRender = out;
        exposureStale = true;
        refiningCounter = 0;
        return out;
    }
    // IImageEvaluator::refine = render(..., refine = true, previous) (image_evaluator_simple.cpp:288-356): a new frame blended into
    // the previous one with weight 1 / refiningCounter (the DVR ray evaluator is deterministic: the average converges at once)
    int refiningCounter = 0;
    torch::Tensor refine(int width, int height, const torch::Tensor& previous) {
        TORCH_CHECK(previous.dim() == 4 && previous.size(0) == (camera ? camera->batches() : 1) && previous.size(1) == 8 && previous.size(2) == height &&
                        previous.size(3) == width, "previous must be a (B,8,H,W) render of the same size");
        const int counter = refiningCounter;
        const torch::Tensor t = render(width, height);
        refiningCounter = counter + 1;
        torch::Tensor out = previous + (t - previous) * (1.0 / refiningCounter);
        lastRender = out;
        exposureStale = true;
        return out;
    }
    py::object moduleForTag(const std::string& tag) const {  // IImageEvaluator::getSelectedModuleForTag
        if (tag == "camera") return py::cast(camera);
        if (tag == "volume") return py::cast(volume);
        if (tag == "RayEvaluation") return py::cast(rayEvaluator);
        auto dvr = std::dynamic_pointer_cast<RayEvaluationSteppingDvr>(rayEvaluator);
        if (dvr && tag == "tf") return py::cast(dvr->tf);
        if (dvr && tag == "brdf") return py::cast(dvr->brdf);
        if (dvr && tag == "blending") return py::cast(dvr->blending);
        raise("no module for tag '" + tag + "' (camera, volume, RayEvaluation, tf, brdf, blending)");
    }
    torch::Tensor lastRender;
    bool exposureStale = false;
    float exposure() {
        // the reference ends render() with a blocking max().item() (image_evaluator_simple.cpp:358); here the
        // device->host sync is deferred until the value is actually needed
        if (exposureStale && lastRender.defined()) {
            lastMaxExposure = lastRender.slice(1, 0, 3).max().item<float>();
            exposureStale = false;
        }
        if (fixMaxExposure) return std::fmax(0.001f, fixedMaxExposure * tonemappingShoulder);
        return std::fmax(0.001f, lastMaxExposure * tonemappingShoulder);
    }

    // IImageEvaluator::ExtractColor (renderer/iimage_evaluator.cpp:26-135): one call into the C ABI per batch entry
    static torch::Tensor extractColorStatic(const torch::Tensor& raw, bool tonemap, float maxExposure, ChannelMode channel) {
        TORCH_CHECK(raw.dim() == 4 && raw.size(1) == 8, "raw input must be of shape (B,8,H,W)");
        TORCH_CHECK(raw.is_cuda() && raw.scalar_type() == torch::kFloat32, "raw input must be a float32 tensor on the GPU");
        const torch::Tensor in = raw.contiguous();
        const int B = int(in.size(0)), H = int(in.size(2)), W = int(in.size(3));
        torch::Tensor out = torch::empty({B, 4, H, W}, in.options());
        for (int b = 0; b < B; ++b)
            check(fvsrn_extract_color(in.data_ptr<float>() + size_t(b) * 8 * H * W, W, H, int(channel), tonemap ? 1 : 0, maxExposure,
                                      out.data_ptr<float>() + size_t(b) * 4 * H * W, currentStream()));
        return out;
    }
};

// -------------------------------------------------------------------------------------------- JSON scene files
// ModuleRegistry::loadTree (renderer/module_registry.cpp:288-305) for the modules of this path; the JSON is parsed
// by Python's json module (plumbing) and walked here.
template <class T> T jget(const py::dict& d, const char* key, T def) {
    if (!d.contains(key)) return def;
    return d[key].cast<T>();
}
py::dict jsub(const py::dict& root, const std::string& tag, const std::string& name) {
    if (!root.contains(tag.c_str())) raise("scene file has no section '" + tag + "'");
    py::dict sec = root[tag.c_str()].cast<py::dict>();
    if (!sec.contains(name.c_str())) raise("scene file has no module '" + tag + "/" + name + "'");
    return sec[name.c_str()].cast<py::dict>();
}

// ---- transfer functions from their JSON description (host-side construction of the device tables) ----------------
struct ColorPoint { double pos; double rgb[3]; };
std::vector<ColorPoint> colorPointsFromJson(const py::dict& jt) {  // adl_serializer<TFPartPiecewiseColor::Point>, transfer_function.h:403-417
    std::vector<ColorPoint> pts;
    for (const auto& v : jt["colorPoints"].cast<std::vector<std::vector<double>>>()) {
        if (v.size() != 4) raise("colorPoints entries must be [position, r, g, b]");
        pts.push_back({v[0], {v[1], v[2], v[3]}});
    }
    if (pts.empty()) raise("the transfer function has no colour control points");
    std::stable_sort(pts.begin(), pts.end(), [](const ColorPoint& a, const ColorPoint& b) { return a.pos < b.pos; });  // sortPointsAndUpdateTexture
    return pts;
}

// TransferFunctionPiecewiseLinear::computeTensor (renderer/transfer_function_piecewise.cpp:166-282): merged control
// points (1,R,5) = [r,g,b,absorption*scaling,pos]
torch::Tensor piecewiseTensor(std::vector<ColorPoint> color, const std::vector<std::vector<double>>& opacityJson, double scaling) {
    struct OP { double pos, absorption; };
    std::vector<OP> opacity;
    for (const auto& v : opacityJson) {
        if (v.size() != 2) raise("opacityPoints entries must be [position, absorption]");
        opacity.push_back({v[0], v[1]});
    }
    if (opacity.empty()) raise("the transfer function has no opacity control points");
    std::stable_sort(opacity.begin(), opacity.end(), [](const OP& a, const OP& b) { return a.pos < b.pos; });
    // control points outside [0,1] (-1 and 2) if the first / last ones are inside (:181-199)
    if (color.front().pos > 0) color.insert(color.begin(), ColorPoint{-1.0, {color.front().rgb[0], color.front().rgb[1], color.front().rgb[2]}});
    if (opacity.front().pos > 0) opacity.insert(opacity.begin(), OP{-1.0, opacity.front().absorption});
    if (color.back().pos < 1) color.push_back(ColorPoint{2.0, {color.back().rgb[0], color.back().rgb[1], color.back().rgb[2]}});
    if (opacity.back().pos < 1) opacity.push_back(OP{2.0, opacity.back().absorption});
    struct P { double pos, rgb[3], absorption; };
    std::vector<P> pts;
    pts.push_back({color[0].pos <= opacity[0].pos ? color[0].pos : opacity[0].pos, {color[0].rgb[0], color[0].rgb[1], color[0].rgb[2]}, opacity[0].absorption});
    size_t io = 0, ic = 0;
    while (io + 1 < opacity.size() && ic + 1 < color.size()) {  // :209-237
        if (opacity[io + 1].pos < color[ic + 1].pos) {
            const double f = (opacity[io + 1].pos - color[ic].pos) / (color[ic + 1].pos - color[ic].pos);
            P q{opacity[io + 1].pos, {0, 0, 0}, opacity[io + 1].absorption};
            for (int k = 0; k < 3; ++k) q.rgb[k] = color[ic].rgb[k] + f * (color[ic + 1].rgb[k] - color[ic].rgb[k]);
            pts.push_back(q);
            ++io;
        } else {
            const double f = (color[ic + 1].pos - opacity[io].pos) / (opacity[io + 1].pos - opacity[io].pos);
            pts.push_back({color[ic + 1].pos, {color[ic + 1].rgb[0], color[ic + 1].rgb[1], color[ic + 1].rgb[2]},
                           opacity[io].absorption + f * (opacity[io + 1].absorption - opacity[io].absorption)});
            ++ic;
        }
    }
    const float EPS = 1e-7f;  // purge runs of zero absorption and coincident points (:243-256)
    for (int64_t i = 0; i < int64_t(pts.size()) - 2;) {
        if ((pts[i].absorption < EPS && pts[i + 1].absorption < EPS && pts[i + 2].absorption < EPS) || (pts[i + 1].pos - pts[i].pos < EPS))
            pts.erase(pts.begin() + (i + 1));
        else
            ++i;
    }
    torch::Tensor t = torch::empty({1, int64_t(pts.size()), 5}, torch::kFloat);
    for (size_t i = 0; i < pts.size(); ++i) {  // clamp colour, scale opacity (:259-263)
        float* r = t.data_ptr<float>() + 5 * i;
        for (int k = 0; k < 3; ++k) r[k] = float(std::min(std::max(pts[i].rgb[k], 0.0), double(1.0f - FLT_EPSILON)));
        r[3] = float(std::min(std::max(pts[i].absorption, 0.0), 1.0) * scaling);
        r[4] = float(pts[i].pos);
    }
    return t;
}

// TransferFunctionTexture::computeTexture (renderer/transfer_function_texture.cpp:347-362) with
// TFPartPiecewiseColor::getAsTexture (transfer_function.cpp:526-551): 256 texels (1,256,4) = [r,g,b,absorptionScaling*plot]
torch::Tensor textureTensor(const std::vector<ColorPoint>& color, const std::vector<float>& plot, double scaling) {
    const int R = 256;
    if (int(plot.size()) != R) raise("opacityPoints of a Texture transfer function must hold 256 values");
    torch::Tensor t = torch::empty({1, R, 4}, torch::kFloat);
    const int n = int(color.size());
    for (int i = 0; i < R; ++i) {
        const float density = (i + 0.5f) / R;
        int idx;
        for (idx = 0; idx < n - 2; ++idx)
            if (color[size_t(idx) + 1].pos > density) break;
        const ColorPoint& lo = color[size_t(idx)];
        const ColorPoint& hi = color[size_t(std::min(idx + 1, n - 1))];
        const float pLow = float(lo.pos), pHigh = float(hi.pos);
        const float frac = std::min(std::max((density - pLow) / (pHigh - pLow), 0.0f), 1.0f);
        float* r = t.data_ptr<float>() + 4 * i;
        for (int k = 0; k < 3; ++k) r[k] = float((1 - frac) * lo.rgb[k] + frac * hi.rgb[k]);
        r[3] = float(scaling) * plot[size_t(i)];
    }
    return t;
}

// RayEvaluationSteppingDvr::convertToTextureTF (ray_evaluation_stepping.cpp:767-779) = TransferFunctionTexture::doPaste
// (transfer_function_texture.cpp:412-437): the current TF is sampled on the host at the 256 texel centres
// (ITransferFunction::evaluate(double): transfer_function_identity.cpp:140-150, _gaussian.cpp:335-338,376-385,
// _piecewise.cpp:283-287 = sampleTF of renderer_tf_piecewise.cuh:31-52), colours become colour control points, absorption the
// opacity plot relative to the scaling (10, raised to the maximum absorption if that is larger), then computeTexture.
void RayEvaluationSteppingDvr::convertToTextureTF() {
    if (!tf) return;
    if (std::dynamic_pointer_cast<TransferFunctionTexture>(tf)) return;  // already a texture TF
    const int R = 256;
    std::vector<ColorPoint> colorPoints;
    std::vector<float> plot(R);
    double scaling = 10.0, maxW = 0.0;
    auto ident = std::dynamic_pointer_cast<TransferFunctionIdentity>(tf);
    auto table = std::dynamic_pointer_cast<TableTF>(tf);
    if (table && !table->tensor.defined()) raise("the transfer function has no control points");
    for (int i = 0; i < R; ++i) {
        const double d = (i + 0.5) / R;
        double rgba[4] = {0, 0, 0, 0};
        if (ident) {
            rgba[0] = rgba[1] = rgba[2] = d * ident->absorptionEmission->value.y;
            rgba[3] = d * ident->absorptionEmission->value.x;
        } else if (table && table->kind == FVSRN_TF_GAUSSIAN) {
            const float* t = table->tensor.data_ptr<float>();
            for (int64_t k = 0; k < table->tensor.size(1); ++k) {
                const float* r = t + 6 * k;
                const float x = float(d);
                const double ni = std::exp(-(x - r[4]) * (x - r[4]) / (r[5] * r[5]));  // gaussian(float, float, float), expf
                for (int c = 0; c < 4; ++c) rgba[c] += double(r[c]) * double(float(ni));
            }
        } else if (table && table->kind == FVSRN_TF_PIECEWISE) {
            const float* t = table->tensor.data_ptr<float>();
            const int n = int(table->tensor.size(1));
            float density = float(d);
            int k;
            for (k = 0; k < n - 2; ++k)
                if (t[5 * (k + 1) + 4] > density) break;
            const float* a = t + 5 * k;
            const float* b = t + 5 * (k + 1);
            density = std::min(std::max(density, a[4]), b[4]);
            const float frac = (density - a[4]) / (b[4] - a[4]);
            for (int c = 0; c < 4; ++c) rgba[c] = a[c] + frac * (b[c] - a[c]);
        } else {
            raise("Copying to texture TF not supported from source TF");
        }
        colorPoints.push_back({d, {rgba[0], rgba[1], rgba[2]}});
        plot[size_t(i)] = float(rgba[3] / scaling);
        maxW = std::max(maxW, rgba[3]);
    }
    if (maxW > scaling) {
        const double f = scaling / maxW;
        scaling = maxW;
        for (float& p : plot) p = float(p * f);
    }
    auto tex = std::make_shared<TransferFunctionTexture>();
    tex->setTensor(textureTensor(colorPoints, plot, scaling));
    tf = tex;
}

std::shared_ptr<ImageEvaluatorSimple> loadFromJson(const std::string& filename) {
    py::object json = py::module_::import("json");
    py::object io = py::module_::import("io");
    py::dict root = json.attr("load")(io.attr("open")(filename, "r")).cast<py::dict>();
    const std::string rootName = jget<std::string>(root, "root", "Simple");
    if (rootName != "Simple") raise("only the 'Simple' image evaluator is supported, the scene file selects '" + rootName + "'");
    py::dict je = jsub(root, "ImageEvaluator", "Simple");
    auto ev = std::make_shared<ImageEvaluatorSimple>();
    ev->sppLog2 = jget<int>(je, "samplesPerIterationLog2", 0);
    ev->useTonemapping = jget<bool>(je, "useTonemapping", false);
    ev->tonemappingShoulder = jget<float>(je, "tonemappingShoulder", 1.f);
    ev->fixMaxExposure = jget<bool>(je, "fixMaxExposure", false);
    ev->fixedMaxExposure = jget<float>(je, "fixedMaxExposure", 1.f);
    // camera (camera.cpp:349-362)
    {
        const std::string sel = jget<std::string>(je, "selectedCamera", "Sphere");
        if (sel != "Sphere") raise("camera '" + sel + "' is not supported (only 'Sphere')");
        py::dict jc = jsub(root, "camera", "Sphere");
        auto cam = std::make_shared<CameraOnASphere>();
        static const char* names[6] = {"Xp", "Xm", "Yp", "Ym", "Zp", "Zm"};
        const std::string o = jget<std::string>(jc, "orientation", "Ym");
        cam->orientation = -1;
        for (int i = 0; i < 6; ++i)
            if (o == names[i]) cam->orientation = i;
        if (cam->orientation < 0) raise("unknown camera orientation " + o);
        if (jc.contains("center")) {
            auto c = jc["center"].cast<std::vector<double>>();
            cam->center->value = {c.at(0), c.at(1), c.at(2)};
        }
        cam->pitchYawDistance->value = {jget<double>(jc, "pitch", 0.0), jget<double>(jc, "yaw", 0.0), jget<double>(jc, "distance", 1.0)};
        cam->fovYRadians = jget<double>(jc, "fovY", cam->fovYRadians);
        ev->camera = cam;
    }
    // ray evaluator (ray_evaluation_stepping.cpp:62-72,740-755)
    {
        const std::string sel = jget<std::string>(je, "selectedRayEvaluator", "DVR");
        if (sel != "DVR") raise("ray evaluator '" + sel + "' is not supported (only 'DVR')");
        py::dict jr = jsub(root, "RayEvaluation", "DVR");
        auto dvr = std::make_shared<RayEvaluationSteppingDvr>();
        double step = jget<double>(jr, "stepsize", 0.005);
        if (jget<bool>(jr, "stepsizeIsObjectSpace", false)) step /= 256.0;
        dvr->stepsize = step;
        dvr->minDensity = jget<double>(jr, "minDensity", 0.0);
        dvr->maxDensity = jget<double>(jr, "maxDensity", 1.0);
        dvr->enableEarlyOut = jget<bool>(jr, "earlyOut", true);
        if (root.contains("blending")) {
            py::dict jb = jsub(root, "blending", "blending");
            dvr->blending->blendMode = jget<std::string>(jb, "blending", "BeerLambert") == "Alpha" ? FVSRN_BLEND_ALPHA : FVSRN_BLEND_BEER_LAMBERT;
        }
        if (root.contains("brdf") && root["brdf"].cast<py::dict>().contains("Lambert")) {
            py::dict jl = jsub(root, "brdf", "Lambert");
            BRDFLambert& b = *dvr->brdf;  // BRDFLambert::load, brdf.cpp:208-225
            b.enableMagnitudeScaling = jget<bool>(jl, "enableMagnitudeScaling", false);
            b.magnitudeScaling->value = jget<double>(jl, "magnitudeScaling", 1.0);
            b.enablePhong = jget<bool>(jl, "enablePhong", false);
            b.ambient->value = jget<double>(jl, "ambient", 1.0);
            b.specular->value = jget<double>(jl, "specular", 1.0);
            b.magnitudeCenter->value = jget<double>(jl, "magnitudeCenter", 1.0);
            b.magnitudeRadius->value = jget<double>(jl, "magnitudeRadius", 1.0);
            b.specularExponent->value = jget<int>(jl, "specularExponent", 1);
            b.lightFollowsCamera = jget<bool>(jl, "lightFollowsCamera", true);
            b.lightType = jget<std::string>(jl, "lightType", "") == "Point" ? FVSRN_LIGHT_POINT : FVSRN_LIGHT_DIRECTIONAL;
            auto vec3 = [&](const char* key) {
                double3 v{0, 0, 0};
                if (jl.contains(key)) {
                    py::list a = jl[key].cast<py::list>();
                    if (py::len(a) == 3) v = double3{a[0].cast<double>(), a[1].cast<double>(), a[2].cast<double>()};
                }
                return v;
            };
            b.lightPosition->value = vec3("lightPosition");
            b.lightDirection->value = vec3("lightDirection");
        }
        const std::string tfSel = jget<std::string>(jr, "selectedTF", "Identity");
        py::dict jt = jsub(root, "tf", tfSel);
        if (tfSel == "Identity") {  // transfer_function_identity.cpp:101-140
            auto tf = std::make_shared<TransferFunctionIdentity>();
            tf->absorptionEmission->value = {jget<double>(jt, "absorptionScaling", 1.0), jget<double>(jt, "emissionScaling", 1.0)};
            dvr->tf = tf;
        } else if (tfSel == "Gaussian") {  // transfer_function_gaussian.cpp:234-242,340-360
            auto tf = std::make_shared<TransferFunctionGaussian>();
            tf->scaleWithGradient = jget<bool>(jt, "scaleWithGradient", false);  // :238-239
            tf->piecewiseAnalyticIntegration = jget<bool>(jt, "usePiecewiseAnalyticIntegration", false);
            const double scale = jget<double>(jt, "absorptionScaling", 1.0);
            auto pts = jt["points"].cast<std::vector<std::vector<double>>>();
            torch::Tensor t = torch::empty({1, int64_t(pts.size()), 6}, torch::kFloat);
            for (size_t i = 0; i < pts.size(); ++i) {
                float* r = t.data_ptr<float>() + 6 * i;
                r[0] = float(pts[i].at(0)); r[1] = float(pts[i].at(1)); r[2] = float(pts[i].at(2));
                r[3] = float(pts[i].at(3) * scale); r[4] = float(pts[i].at(4)); r[5] = float(pts[i].at(5));
            }
            tf->setTensor(t);
            dvr->tf = tf;
        } else if (tfSel == "Piecewise") {
            auto tf = std::make_shared<TransferFunctionPiecewise>();
            tf->setTensor(piecewiseTensor(colorPointsFromJson(jt), jt["opacityPoints"].cast<std::vector<std::vector<double>>>(),
                                          jget<double>(jt, "absorptionScaling", 1.0)));
            dvr->tf = tf;
        } else if (tfSel == "Texture") {
            const std::string pre = jget<std::string>(jt, "preintegrationMode", "None");  // magic_enum names, transfer_function_texture.cpp:222
            auto tf = std::make_shared<TransferFunctionTexture>();
            if (pre == "Preintegrate1D") tf->preintegrationMode = FVSRN_PREINTEGRATE_1D;
            else if (pre == "Preintegrate2D") tf->preintegrationMode = FVSRN_PREINTEGRATE_2D;
            tf->setTensor(textureTensor(colorPointsFromJson(jt), jt["opacityPoints"].cast<std::vector<float>>(),
                                        jget<double>(jt, "absorptionScaling", 1.0)));
            dvr->tf = tf;
        } else {
            raise("unknown transfer function '" + tfSel + "' (Identity, Gaussian, Piecewise, Texture)");
        }
        ev->rayEvaluator = dvr;
    }
    // volume: networks are not stored in the JSON (volume_interpolation_network.cpp:1664-1672)
    {
        // The reference's scene files select the ground-truth volume ("Grid", "Implicit"); its callers usually replace it by
        // the trained network (inference.py:598 `image_evaluator.volume = self._volume_network`).  "Grid" is loaded like
        // VolumeInterpolationGrid::load (volume_interpolation_grid.cpp:772-812): the .cvol next to the scene file if it
        // exists, otherwise an empty source; "Implicit" is recorded only.
        const std::string sel = jget<std::string>(je, "selectedVolume", "SRN");
        if (sel == "SRN") {
            ev->volume = std::make_shared<VolumeInterpolationNetwork>();
        } else if (sel == "Grid") {
            auto grid = std::make_shared<VolumeInterpolationGrid>();
            if (root.contains("volume") && root["volume"].cast<py::dict>().contains("Grid")) {
                py::dict jg = jsub(root, "volume", "Grid");
                const std::string ip = jget<std::string>(jg, "interpolation", "TRILINEAR");
                grid->interpolation_ = ip == "NEAREST_NEIGHBOR" ? FVSRN_VOLUME_NEAREST : (ip == "TRICUBIC" ? FVSRN_VOLUME_TRICUBIC : FVSRN_VOLUME_TRILINEAR);
                const std::string src = jget<std::string>(jg, "source", "");
                std::string path = jget<std::string>(jg, "volumePath", "");
                if (src == "VOLUME" && !path.empty() && jget<int>(jg, "mipmapLevel", 0) == 0) {
                    py::object os = py::module_::import("os.path");
                    if (!os.attr("isabs")(path).cast<bool>())
                        path = os.attr("normpath")(os.attr("join")(os.attr("dirname")(os.attr("abspath")(filename)), path)).cast<std::string>();
                    if (path.size() > 5 && path.substr(path.size() - 5) == ".cvol" && os.attr("exists")(path).cast<bool>()) {
                        try {
                            grid->setSourceVolume(std::make_shared<Volume>(path), 0);
                        } catch (const std::exception& e) {  // the reference prints and leaves the source empty
                            py::print("Unable to load volume, is the file valid?", path, e.what());
                        }
                    }
                }
            }
            ev->volume = grid;
        } else {
            ev->volume = std::make_shared<UnsupportedVolume>(sel);
        }
    }
    return ev;
}

}  // namespace

PYBIND11_MODULE(pyrenderer, m) {
    m.doc() = "MI355X-native drop-in for the SRN/DVR path of fV-SRN's pyrenderer";
    py::module_::import("torch");  // the Tensor type casters need torch's Python side, whatever the import order of the caller
    // bindings/bindings.cpp:155-171: cache configuration is meaningless for ahead-of-time kernels; kept as no-ops
    m.def("set_cuda_cache_dir", [](const std::string&) {});
    m.def("set_kernel_cache_file", [](const std::string&) {});
    m.def("disable_cuda_cache", []() {});
    m.def("cleanup", []() {});
    m.def("sync", []() { py::module_::import("torch").attr("cuda").attr("synchronize")(); });
    m.def("version", []() { return std::string(fvsrn_version()); });

    bindV3<float>(m, "float3"); bindV3<double>(m, "double3"); bindV3<int>(m, "int3");
    bindV4<float>(m, "float4"); bindV4<double>(m, "double4");
    py::class_<int2>(m, "int2").def(py::init<>()).def(py::init([](int x, int y) { return int2{x, y}; }))
        .def_readwrite("x", &int2::x).def_readwrite("y", &int2::y);
    py::class_<V2<double>>(m, "double2").def(py::init<>()).def(py::init([](double x, double y) { return V2<double>{x, y}; }))
        .def_readwrite("x", &V2<double>::x).def_readwrite("y", &V2<double>::y);
    bindParameter<double3>(m, "Parameter_double3");
    bindParameter<V2<double>>(m, "Parameter_double2");
    bindParameter<double>(m, "Parameter_double");
    bindParameter<int>(m, "Parameter_int");

    py::class_<GPUTimer>(m, "GPUTimer").def(py::init<>()).def("start", &GPUTimer::start).def("stop", &GPUTimer::stop)
        .def("elapsed_milliseconds", &GPUTimer::elapsed);

    // ---- SceneNetwork and nested classes (volume_interpolation_network.cpp:1828-1972)
    py::class_<SceneNetwork, std::shared_ptr<SceneNetwork>> sn(m, "SceneNetwork");
    py::class_<InputParametrization, std::shared_ptr<InputParametrization>>(sn, "InputParametrization")
        .def_property("has_time", [](InputParametrization& p) { return p.hasTime; }, [](InputParametrization& p, bool v) { p.hasTime = v; p.push(); })
        .def_property("has_direction", [](InputParametrization& p) { return p.hasDirection; }, [](InputParametrization& p, bool v) { p.hasDirection = v; p.push(); })
        .def("num_fourier_features", [](InputParametrization& p) { return p.numFourier; })
        .def("set_fourier_matrix_from_tensor", [](InputParametrization& p, const torch::Tensor& t, bool premultiplied) {
            TORCH_CHECK(t.dim() == 2, "the fourier matrix must be of shape (F,3) or (F,6)");
            const int cols = int(t.size(1));
            if (cols == 6 && !p.hasDirection) raise("hasDirection==false, but the fourier matrix has input channels for the direction");
            if (cols != 3 && cols != 6) raise("Unrecognized number of input channels. Actual: " + std::to_string(cols) + ", expected: 3 or 6");
            torch::Tensor c = t.detach().to(c10::kCPU, c10::kFloat).contiguous();
            p.fourier.assign(c.data_ptr<float>(), c.data_ptr<float>() + c.numel());
            p.numFourier = int(t.size(0));
            p.useDirInFourier = cols == 6;
            p.premultiplied = premultiplied;
            p.push();
        })
        .def("disable_fourier_features", [](InputParametrization& p) { p.numFourier = 0; p.useDirInFourier = false; p.fourier.clear(); p.push(); })
        .def("channels_out", &InputParametrization::channelsOut)
        .def("valid", [](InputParametrization& p) { return !(p.useDirInFourier && !p.hasDirection) && p.numFourier % 2 == 0; });

    py::class_<OutputParametrization, std::shared_ptr<OutputParametrization>> om(sn, "OutputParametrization");
    py::enum_<fvsrn_output_mode>(om, "OutputMode")
        .value("DENSITY", FVSRN_OUT_DENSITY).value("DENSITY_DIRECT", FVSRN_OUT_DENSITY_DIRECT)
        .value("RGBO", FVSRN_OUT_RGBO).value("RGBO_DIRECT", FVSRN_OUT_RGBO_DIRECT)
        .value("DENSITY_GRADIENT", FVSRN_OUT_DENSITY_GRADIENT).value("DENSITY_GRADIENT_DIRECT", FVSRN_OUT_DENSITY_GRADIENT_DIRECT)
        .value("DENSITY_GRADIENT_CUBIC", FVSRN_OUT_DENSITY_GRADIENT_CUBIC)
        .value("DENSITY_CURVATURE", FVSRN_OUT_DENSITY_CURVATURE).value("DENSITY_CURVATURE_DIRECT", FVSRN_OUT_DENSITY_CURVATURE_DIRECT)
        .export_values();
    om.def_property("output_mode", [](OutputParametrization& o) { return o.mode; }, [](OutputParametrization& o, fvsrn_output_mode v) { o.mode = v; o.push(); })
        .def_static("OutputModeFromString", [](const std::string& s) {
            static const char* names[9] = {"density", "density:direct", "rgbo", "rgbo:direct", "densitygrad", "densitygrad:direct",
                                           "densitygrad:cubic", "densitycurvature", "densitycurvature:direct"};
            for (int i = 0; i < 9; ++i)
                if (s == names[i]) return fvsrn_output_mode(i);
            raise("No output mode found matching string " + s);
        })
        .def("channels_in", [](OutputParametrization& o) { static const int c[9] = {1, 1, 4, 4, 4, 4, 4, 6, 6}; return c[int(o.mode)]; });

    py::class_<Layer, std::shared_ptr<Layer>> l(sn, "Layer");
    py::enum_<fvsrn_activation>(l, "Activation")
        .value("ReLU", FVSRN_ACT_RELU).value("Sine", FVSRN_ACT_SINE).value("Snake", FVSRN_ACT_SNAKE)
        .value("SnakeAlt", FVSRN_ACT_SNAKEALT).value("Sigmoid", FVSRN_ACT_SIGMOID).value("NONE", FVSRN_ACT_NONE)
        .export_values();
    l.def_readonly("activation", &Layer::activation)
        .def_static("ActivationFromString", [](const std::string& s) {
            static const char* names[6] = {"ReLU", "Sine", "Snake", "SnakeAlt", "Sigmoid", "None"};
            for (int i = 0; i < 6; ++i)
                if (s == names[i]) return fvsrn_activation(i);
            raise("No output mode found matching string " + s);
        })
        .def("valid", [](const Layer& l, bool isOutputLayer) {  // Layer::valid, volume_interpolation_network.cpp:241-246 (sizes always match here)
                 return l.channelsIn > 0 && l.channelsOut > 0 && (isOutputLayer || l.channelsOut % 4 == 0);
             }, py::arg("is_output_layer"))
        .def_readonly("channels_in", &Layer::channelsIn)
        .def_readonly("channels_out", &Layer::channelsOut);

    py::class_<LatentGrid, std::shared_ptr<LatentGrid>> lg(sn, "LatentGrid");
    py::enum_<fvsrn_grid_encoding>(lg, "Encoding")
        .value("Float", FVSRN_GRID_FLOAT).value("ByteLinear", FVSRN_GRID_BYTE_LINEAR).value("ByteGaussian", FVSRN_GRID_BYTE_GAUSSIAN)
        .export_values();
    lg.def(py::init<>())
        .def(py::init(&LatentGrid::fromTensor))
        .def("is_valid", &LatentGrid::isValid)
        .def_readonly("grid_channels", &LatentGrid::gridChannels)
        .def_readonly("grid_size_z", &LatentGrid::gridSizeZ)
        .def_readonly("grid_size_y", &LatentGrid::gridSizeY)
        .def_readonly("grid_size_x", &LatentGrid::gridSizeX)
        .def_readonly("encoding", &LatentGrid::encoding);

    py::class_<LatentGridTimeAndEnsemble, std::shared_ptr<LatentGridTimeAndEnsemble>>(sn, "LatentGridTimeAndEnsemble")
        .def(py::init<>())
        .def(py::init<int, int, int, int, int>(), py::arg("time_min"), py::arg("time_num"), py::arg("time_step"),
             py::arg("ensemble_min"), py::arg("ensemble_num"))
        .def_readonly("time_min", &LatentGridTimeAndEnsemble::timeMin)
        .def_readonly("time_num", &LatentGridTimeAndEnsemble::timeNum)
        .def_readonly("time_step", &LatentGridTimeAndEnsemble::timeStep)
        .def_readonly("ensemble_min", &LatentGridTimeAndEnsemble::ensembleMin)
        .def_readonly("ensemble_num", &LatentGridTimeAndEnsemble::ensembleNum)
        .def_property_readonly("time_max_inclusive", &LatentGridTimeAndEnsemble::timeMaxInclusive)
        .def_property_readonly("ensemble_max_inclusive", &LatentGridTimeAndEnsemble::ensembleMaxInclusive)
        .def("interpolate_time", &LatentGridTimeAndEnsemble::interpolateTime, py::arg("time"))
        .def("interpolate_ensemble", &LatentGridTimeAndEnsemble::interpolateEnsemble, py::arg("time"))
        .def("get_time_grid", [](LatentGridTimeAndEnsemble& g, int i) { TORCH_CHECK(i >= 0 && i < g.timeNum, "Index out of bounds"); return g.timeGrids[size_t(i)]; }, py::arg("index"))
        .def("get_ensemble_grid", [](LatentGridTimeAndEnsemble& g, int i) { TORCH_CHECK(i >= 0 && i < g.ensembleNum, "Index out of bounds"); return g.ensembleGrids[size_t(i)]; }, py::arg("index"))
        .def("set_time_grid_from_torch", [](LatentGridTimeAndEnsemble& g, int i, const torch::Tensor& t, fvsrn_grid_encoding e) { return g.setGrid(false, i, t, e); },
             py::arg("index"), py::arg("tensor"), py::arg("encoding"))
        .def("set_ensemble_grid_from_torch", [](LatentGridTimeAndEnsemble& g, int i, const torch::Tensor& t, fvsrn_grid_encoding e) { return g.setGrid(true, i, t, e); },
             py::arg("index"), py::arg("tensor"), py::arg("encoding"))
        .def("is_valid", &LatentGridTimeAndEnsemble::isValid)
        .def("common_encoding", &LatentGridTimeAndEnsemble::commonEncoding)
        .def("time_channels", &LatentGridTimeAndEnsemble::timeChannels)
        .def("ensemble_channels", &LatentGridTimeAndEnsemble::ensembleChannels);

    sn.def(py::init(&SceneNetwork::create))
        .def_property_readonly("input", [](SceneNetwork& n) { return n.input; })
        .def_property_readonly("output", [](SceneNetwork& n) { return n.output; })
        .def_property("latent_grid", [](SceneNetwork& n) { return n.latentGrid; }, &SceneNetwork::setLatentGrid)
        .def("add_layer", &SceneNetwork::addLayer, py::arg("weights"), py::arg("bias"), py::arg("activation"),
             py::arg("activation_parameter") = 1.0f)
        .def("num_layers", &SceneNetwork::numLayers)
        .def("get_layer", &SceneNetwork::getLayer)
        .def_property("box_min", [](SceneNetwork& n) { return n.boxMin; }, [](SceneNetwork& n, float3 v) { n.setBox(v, n.boxSize); })
        .def_property("box_size", [](SceneNetwork& n) { return n.boxSize; }, [](SceneNetwork& n, float3 v) { n.setBox(n.boxMin, v); })
        .def("valid", &SceneNetwork::valid)
        .def("save", &SceneNetwork::save)
        .def_static("load", &SceneNetwork::load)
        .def("num_parameters", [](SceneNetwork& n) { return n.info().num_parameters; })
        .def("compute_max_warps", [](SceneNetwork& n, bool onlyShared, bool adjoint) {
            // SceneNetwork::computeMaxWarps(.., adjoint) :987-1041: the reference's shared-memory budget with one more C-wide
            // row of halfs per layer and thread; informational here (the HIP path keeps no activation store, srn_gradient.hpp)
            const fvsrn_network_info i = n.info();
            const int plain = onlyShared ? i.max_warps_shared : i.max_warps_mixed;
            if (!adjoint || plain <= 0) return plain;
            const int maxCh = std::max(i.hidden_channels + i.grid_channels, 1);
            const int freeBytes = plain * maxCh * 2 * 32;  // (48 KiB - weights) rounded down to whole warps
            const int w = freeBytes / (maxCh * i.num_layers * 2 * 32);
            return w > 0 ? w : -1;
        }, py::arg("only_shared_memory"), py::arg("adjoint") = false)
        .def("clear_gpu_resources", [](SceneNetwork& n) { check(fvsrn_network_clear_gpu_resources(n.h)); })
        .def("set_time_and_ensemble", [](SceneNetwork& n, float t, int e) { check(fvsrn_network_set_time_and_ensemble(n.h, t, e)); },
             py::arg("time"), py::arg("ensemble"))
        .def("flops_per_sample", [](SceneNetwork& n) { return n.info().flops_per_sample; });

    // ---- volumes (volume_interpolation.cpp:615-695, volume_interpolation_network.cpp:1974-2005)
    py::class_<IVolumeInterpolation, std::shared_ptr<IVolumeInterpolation>>(m, "IVolumeInterpolation")
        .def("box_min", [](IVolumeInterpolation& v) { return v.boxMin_; })
        .def("box_max", [](IVolumeInterpolation& v) { return v.boxMax_; })
        .def("box_size", &IVolumeInterpolation::boxSize)
        .def("set_box_min", &IVolumeInterpolation::setBoxMin)
        .def("set_box_max", &IVolumeInterpolation::setBoxMax)
        .def("output_channels", &IVolumeInterpolation::outputChannels)
        .def("voxel_size", &IVolumeInterpolation::voxelSize)
        .def("object_resolution", [](IVolumeInterpolation& v) { return int3{v.objectResolution_[0], v.objectResolution_[1], v.objectResolution_[2]}; })
        .def("evaluate", &IVolumeInterpolation::evaluate, py::arg("positions"), py::arg("direction") = std::optional<torch::Tensor>{},
             py::doc("Evaluates the volume on the given position array of shape (B,3) and returns the interpolated densities of shape (B,1)"))
        .def("evaluate_with_gradients", &IVolumeInterpolation::evaluateWithGradients, py::arg("positions"), py::arg("direction") = std::optional<torch::Tensor>{},
             py::doc("Densities of shape (B,1) and gradients of shape (B,3)"))
        .def("evaluate_with_gradients_and_curvature", &IVolumeInterpolation::evaluateWithGradientsAndCurvature, py::arg("positions"),
             py::arg("direction") = std::optional<torch::Tensor>{}, py::doc("Densities (B,1), gradients (B,3) and curvature values (B,2)"))
        // volume_interpolation.cpp:651-695: training-data samplers, not on the inference path (out of scope): present, and raise
        .def("importance_sampling", [](IVolumeInterpolation&, py::args, py::kwargs) -> py::object {
                 raise("importance_sampling is not part of this build (training-data sampler, outside the inference hot path)");
             })
        .def("importance_sampling_with_probability_grid", [](IVolumeInterpolation&, py::args, py::kwargs) -> py::object {
                 raise("importance_sampling_with_probability_grid is not part of this build (training-data sampler, outside the inference hot path)");
             });
    py::class_<UnsupportedVolume, IVolumeInterpolation, std::shared_ptr<UnsupportedVolume>>(m, "UnsupportedVolume")
        .def_readonly("name", &UnsupportedVolume::name);
    // ---- grid volumes (volume.cpp:1244-1400, volume_interpolation_grid.cpp:851-897)
    py::class_<Volume, std::shared_ptr<Volume>> vol(m, "Volume");
    py::enum_<Volume::DataType>(vol, "DataType")
        .value("TypeUChar", Volume::TypeUChar).value("TypeUShort", Volume::TypeUShort).value("TypeFloat", Volume::TypeFloat);
    vol.def_static("bytes_per_type", [](Volume::DataType t) { return int(Volume::bytesPerType(t)); });
    py::class_<Volume::Feature, std::shared_ptr<Volume::Feature>>(vol, "Feature")
        .def("name", &Volume::Feature::name)
        .def("type", &Volume::Feature::type)
        .def("channels", &Volume::Feature::channels)
        .def("base_resolution", &Volume::Feature::baseResolution, py::doc("The resolution of mipmap level 0"));
    vol.def(py::init<>(), py::doc("Creates a new, empty volume"))
        .def(py::init<const std::string&>(), py::doc("Loads the volume from the given .cvol file"))
        .def("save", &Volume::save, py::arg("filename"), py::arg("compression") = 0)
        .def_readwrite("worldX", &Volume::worldX)
        .def_readwrite("worldY", &Volume::worldY)
        .def_readwrite("worldZ", &Volume::worldZ)
        .def("num_features", [](const Volume& v) { return int(v.features.size()); })
        .def("get_feature", &Volume::getFeature, py::arg("index"))
        .def("get_feature", &Volume::getFeatureByName, py::arg("name"))
        .def("add_feature_from_tensor", &Volume::addFeatureFromTensor, py::arg("name"), py::arg("buffer"),
             py::doc("Adds a new feature with the given name from the given CPU-float tensor, a 4D array with the dimensions Channel,X,Y,Z."));
    py::class_<VolumeInterpolationGrid, IVolumeInterpolation, std::shared_ptr<VolumeInterpolationGrid>> vg(m, "VolumeInterpolationGrid");
    py::enum_<VolumeInterpolationGrid::VolumeSource>(vg, "VolumeSource")
        .value("Volume", VolumeInterpolationGrid::SourceVolume).value("TorchTensor", VolumeInterpolationGrid::SourceTorchTensor)
        .value("Empty", VolumeInterpolationGrid::SourceEmpty).export_values();
    py::enum_<fvsrn_volume_interpolation>(vg, "VolumeInterpolation")
        .value("NearestNeighbor", FVSRN_VOLUME_NEAREST).value("Trilinear", FVSRN_VOLUME_TRILINEAR).value("Tricubic", FVSRN_VOLUME_TRICUBIC)
        .export_values();
    vg.def(py::init<>())
        .def("source", [](VolumeInterpolationGrid& g) { return g.source_; })
        .def("interpolation", [](VolumeInterpolationGrid& g) { return fvsrn_volume_interpolation(g.interpolation_); })
        .def("setInterpolation", [](VolumeInterpolationGrid& g, fvsrn_volume_interpolation i) { g.interpolation_ = int(i); })
        .def("minDensity", [](VolumeInterpolationGrid& g) { return g.minDensity_; })
        .def("maxDensity", [](VolumeInterpolationGrid& g) { return g.maxDensity_; })
        .def("volume", [](VolumeInterpolationGrid& g) { return g.volume_; })
        .def("mipmap_level", [](VolumeInterpolationGrid& g) { return g.mipmapLevel_; })
        .def("tensor", [](VolumeInterpolationGrid& g) { return g.tensor_; })
        .def("setSource", &VolumeInterpolationGrid::setSourceVolume, py::arg("volume"), py::arg("mipmap") = 0)
        .def("setSource", &VolumeInterpolationGrid::setSourceTensor, py::arg("tensor"))
        .def_readwrite("grid_resolution_new_behavior", &VolumeInterpolationGrid::newBehavior);
    py::class_<VolumeInterpolationNetwork, IVolumeInterpolation, std::shared_ptr<VolumeInterpolationNetwork>> vn(m, "VolumeInterpolationNetwork");
    py::enum_<GradientMode>(vn, "GradientMode")
        .value("OFF_OR_DIRECT", OFF_OR_DIRECT).value("FINITE_DIFFERENCES", FINITE_DIFFERENCES).value("ADJOINT_METHOD", ADJOINT_METHOD);
    vn.def(py::init<>())
        .def("set_network", &VolumeInterpolationNetwork::setNetwork)
        .def("current_network", &VolumeInterpolationNetwork::currentNetwork)
        .def("set_time_and_ensemble", &VolumeInterpolationNetwork::setTimeAndEnsemble)
        .def_readwrite("only_shared_memory", &VolumeInterpolationNetwork::onlySharedMemory)
        .def_property("gradient_mode", [](VolumeInterpolationNetwork& v) { return v.gradientMode; },
                      [](VolumeInterpolationNetwork& v, GradientMode mode) { v.gradientMode = mode; })
        .def_readwrite("finite_differences_stepsize", &VolumeInterpolationNetwork::finiteDifferencesStepsize)
        .def_readwrite("adjoint_latent_grid_central_differences_stepsize_scale", &VolumeInterpolationNetwork::adjointScale);

    // ---- transfer functions, BRDF, blending
    py::class_<ITransferFunction, std::shared_ptr<ITransferFunction>>(m, "ITransferFunction")
        .def("evaluate", [](ITransferFunction& t, const torch::Tensor& densities, double mn, double mx, const std::optional<torch::Tensor>& gradients) {
                 if (gradients) raise("transfer functions that use the gradient are not in the compiled variant set");
                 return t.evaluate(densities, mn, mx, std::nullopt, 1.0);
             }, py::arg("densities"), py::arg("min_density"), py::arg("max_density"), py::arg("gradients") = std::optional<torch::Tensor>())
        .def("evaluate_with_previous", [](ITransferFunction& t, const torch::Tensor& densities, double mn, double mx, const torch::Tensor& previous,
                                          double stepsize, const std::optional<torch::Tensor>& gradients) {
                 if (gradients) raise("transfer functions that use the gradient are not in the compiled variant set");
                 return t.evaluate(densities, mn, mx, previous, stepsize);
             }, py::arg("densities"), py::arg("min_density"), py::arg("max_density"), py::arg("previous_density"), py::arg("stepsize"),
             py::arg("gradients") = std::optional<torch::Tensor>())
        .def("requires_gradients", [](ITransferFunction& t) {  // ITransferFunction::requiresGradients (transfer_function.cpp:216-223):
                 auto g = dynamic_cast<TransferFunctionGaussian*>(&t);  // only the Gaussian TF with scaleWithGradient asks for normals
                 return g != nullptr && g->scaleWithGradient;
             })
        .def("get_max_absorption", [](ITransferFunction& t) {  // ITransferFunction::getMaxAbsorption: per unit step size
                 fvsrn_scene_desc d{};
                 std::vector<float> table;
                 t.fill(d, table);
                 if (d.tf_kind == FVSRN_TF_IDENTITY) return double(d.tf_scale_absorption);
                 const int cols = d.tf_kind == FVSRN_TF_GAUSSIAN ? 6 : (d.tf_kind == FVSRN_TF_PIECEWISE ? 5 : 4);
                 double m = 0;  // the table rows carry absorption * absorptionScaling in column 3
                 for (size_t i = 3; i < table.size(); i += size_t(cols)) m = std::max(m, double(table[i]));
                 return m;
             });
    py::class_<TransferFunctionIdentity, ITransferFunction, std::shared_ptr<TransferFunctionIdentity>>(m, "TransferFunctionIdentity")
        .def(py::init<>())
        .def_readonly("absorption_emission", &TransferFunctionIdentity::absorptionEmission);
    py::class_<TransferFunctionGaussian, ITransferFunction, std::shared_ptr<TransferFunctionGaussian>>(m, "TransferFunctionGaussian")
        .def(py::init<>())
        .def_property("tensor", [](TransferFunctionGaussian& t) { return t.tensor; }, &TransferFunctionGaussian::setTensor)
        .def_readwrite("absorption_scaling", &TransferFunctionGaussian::absorptionScaling)
        .def_readwrite("piecewise_analytic_integraton", &TransferFunctionGaussian::piecewiseAnalyticIntegration)  // (sic: the reference's spelling)
        .def_readwrite("scale_with_gradient", &TransferFunctionGaussian::scaleWithGradient);
    py::class_<TransferFunctionPiecewise, ITransferFunction, std::shared_ptr<TransferFunctionPiecewise>>(m, "TransferFunctionPiecewiseLinear")
        .def(py::init<>())
        .def_property("tensor", [](TransferFunctionPiecewise& t) { return t.tensor; }, &TransferFunctionPiecewise::setTensor);
    m.attr("TransferFunctionPiecewise") = m.attr("TransferFunctionPiecewiseLinear");  // earlier name of this build
    py::class_<TransferFunctionTexture, ITransferFunction, std::shared_ptr<TransferFunctionTexture>> ttex(m, "TransferFunctionTexture");
    py::enum_<fvsrn_tf_preintegration>(ttex, "PreintegrationMode")  // transfer_function_texture.cpp:246-250
        .value("Off", FVSRN_PREINTEGRATE_NONE).value("Preintegrate1D", FVSRN_PREINTEGRATE_1D).value("Preintegrate2D", FVSRN_PREINTEGRATE_2D).export_values();
    ttex.def(py::init<>())
        .def_property("preintegration_mode", [](TransferFunctionTexture& t) { return fvsrn_tf_preintegration(t.preintegrationMode); },
                      [](TransferFunctionTexture& t, fvsrn_tf_preintegration v) { t.preintegrationMode = v; })
        .def_property("tensor", [](TransferFunctionTexture& t) { return t.tensor; }, &TransferFunctionTexture::setTensor);
    py::class_<IBRDF, std::shared_ptr<IBRDF>>(m, "IBRDF");  // brdf.cpp:107 (its tensor evaluate() is not part of this build)
    py::class_<BRDFLambert, IBRDF, std::shared_ptr<BRDFLambert>> bc(m, "BRDFLambert");  // brdf.cpp:256-273
    py::enum_<fvsrn_light_type>(bc, "LightType").value("Point", FVSRN_LIGHT_POINT).value("Directional", FVSRN_LIGHT_DIRECTIONAL).export_values();
    bc.def(py::init<>())
        .def_readwrite("enable_phong", &BRDFLambert::enablePhong)
        .def_readwrite("enable_magnitude_scaling", &BRDFLambert::enableMagnitudeScaling)
        .def_readonly("magnitude_scaling", &BRDFLambert::magnitudeScaling)
        .def_readonly("ambient", &BRDFLambert::ambient)
        .def_readonly("specular", &BRDFLambert::specular)
        .def_readonly("magnitude_center", &BRDFLambert::magnitudeCenter)
        .def_readonly("magnitude_radius", &BRDFLambert::magnitudeRadius)
        .def_readonly("specular_exponent", &BRDFLambert::specularExponent)
        .def_readwrite("light_follows_camera", &BRDFLambert::lightFollowsCamera)
        .def_property("light_type", [](BRDFLambert& b) { return fvsrn_light_type(b.lightType); }, [](BRDFLambert& b, fvsrn_light_type t) { b.lightType = t; })
        .def_readonly("light_position", &BRDFLambert::lightPosition)
        .def_readonly("light_direction", &BRDFLambert::lightDirection);
    py::class_<Blending, std::shared_ptr<Blending>> bl(m, "Blending");
    py::enum_<fvsrn_blend_mode>(bl, "BlendMode").value("Alpha", FVSRN_BLEND_ALPHA).value("BeerLambert", FVSRN_BLEND_BEER_LAMBERT).export_values();
    bl.def(py::init<>())
        .def_property("blendMode", [](Blending& b) { return fvsrn_blend_mode(b.blendMode); }, [](Blending& b, fvsrn_blend_mode v) { b.blendMode = v; });

    // ---- ray evaluators (ray_evaluation_stepping.cpp:80-93,781-801)
    py::class_<IRayEvaluation, std::shared_ptr<IRayEvaluation>>(m, "IRayEvaluation", py::dynamic_attr());
    py::class_<IRayEvaluationStepping, IRayEvaluation, std::shared_ptr<IRayEvaluationStepping>>(m, "IRayEvaluationStepping")
        .def_readwrite("stepsize", &IRayEvaluationStepping::stepsize);
    py::class_<RayEvaluationSteppingDvr, IRayEvaluationStepping, std::shared_ptr<RayEvaluationSteppingDvr>>(m, "RayEvaluationSteppingDvr")
        .def(py::init<>())
        .def_readwrite("min_density", &RayEvaluationSteppingDvr::minDensity)
        .def_readwrite("max_density", &RayEvaluationSteppingDvr::maxDensity)
        .def_readwrite("early_out", &RayEvaluationSteppingDvr::enableEarlyOut)
        .def_readonly("blending", &RayEvaluationSteppingDvr::blending)
        .def_readwrite("tf", &RayEvaluationSteppingDvr::tf)
        .def_readwrite("brdf", &RayEvaluationSteppingDvr::brdf)
        .def("convert_to_texture_tf", &RayEvaluationSteppingDvr::convertToTextureTF);

    // ---- cameras (camera.cpp:184-224,375-397)
    py::class_<ICamera, std::shared_ptr<ICamera>>(m, "ICamera")
        .def_readonly("aspect_ratio", &ICamera::aspectRatio)
        .def_readwrite("fov_y_radians", &ICamera::fovYRadians)
        // camera.cpp:166-182: projects through the OpenGL matrices of the rasterisation path, which is not part of this build
        .def("world2screen", [](ICamera&, py::args, py::kwargs) -> py::object {
                 raise("world2screen is not part of this build (OpenGL view / projection matrices of the rasterisation path)");
             });
    py::class_<CameraOnASphere, ICamera, std::shared_ptr<CameraOnASphere>> cs(m, "CameraOnASphere");
    py::enum_<Orientation>(cs, "Orientation")
        .value("Xp", Xp).value("Xm", Xm).value("Yp", Yp).value("Ym", Ym).value("Zp", Zp).value("Zm", Zm).export_values();
    cs.def(py::init<>())
        .def_property("orientation", [](CameraOnASphere& c) { return Orientation(c.orientation); },
                      [](CameraOnASphere& c, Orientation o) { c.orientation = int(o); })
        .def_readonly("center", &CameraOnASphere::center)
        .def_readonly("pitchYawDistance", &CameraOnASphere::pitchYawDistance)
        .def("get_origin", &CameraOnASphere::getOrigin, py::arg("batch") = 0)
        .def("get_front", &CameraOnASphere::getFront, py::arg("batch") = 0)
        .def("get_parameters", &CameraOnASphere::getParameters)
        .def("set_parameters", &CameraOnASphere::setParameters)
        .def("generate_rays", [](CameraOnASphere& c, int width, int height, bool doublePrecision) {  // camera.cpp:204-208
                 if (doublePrecision) raise("double precision rays are not supported by this build");
                 c.aspectRatio = double(width) / height;
                 float e[3], r[3], u[3];
                 c.frame(e, r, u);
                 auto opt = torch::TensorOptions().dtype(torch::kFloat).device(torch::kCUDA);
                 torch::Tensor start = torch::empty({1, height, width, 3}, opt), dir = torch::empty({1, height, width, 3}, opt);
                 check(fvsrn_generate_rays(e, r, u, float(c.fovYRadians), width, height, start.data_ptr<float>(), dir.data_ptr<float>(), currentStream()));
                 return std::make_tuple(start, dir);
             }, py::arg("width"), py::arg("height"), py::arg("double_precision") = false)
        // ICamera::generateRaysMultisampling (camera.cpp:100-164, CameraGenerateRayMultisamplingKernel): num_samples rays per pixel
        // through uniformly jittered positions inside the pixel, batch dimension = sample.  The reference seeds its device
        // sampler with (42, time); here torch's generator is seeded with 42 + time: same distribution, another sequence.
        .def("generate_rays_multisampling", [](CameraOnASphere& c, int width, int height, int numSamples, unsigned time, bool doublePrecision) {
                 if (doublePrecision) raise("double precision rays are not supported by this build");
                 if (numSamples <= 0) raise("num_samples must be positive");
                 c.aspectRatio = double(width) / height;
                 float e[3], r[3], u[3];
                 c.frame(e, r, u);
                 auto opt = torch::TensorOptions().dtype(torch::kFloat).device(torch::kCUDA);
                 auto gen = at::detail::createCPUGenerator(42 + uint64_t(time));
                 const torch::Tensor jitter = (torch::rand({numSamples, height, width, 2}, gen, torch::TensorOptions().dtype(torch::kFloat)) - 0.5).to(torch::kCUDA);
                 const torch::Tensor xs = torch::arange(width, opt).view({1, 1, width}), ys = torch::arange(height, opt).view({1, height, 1});
                 const torch::Tensor ndcx = 2 * (xs + jitter.select(3, 0) + 0.5) / width - 1, ndcy = 2 * (ys + jitter.select(3, 1) + 0.5) / height - 1;
                 const torch::Tensor E = torch::tensor({e[0], e[1], e[2]}, opt), R = torch::tensor({r[0], r[1], r[2]}, opt), U = torch::tensor({u[0], u[1], u[2]}, opt);
                 const torch::Tensor F = torch::cross(U, R, 0);  // front = cross(up, right), renderer_camera.cuh:47
                 const float tanY = std::tan(float(c.fovYRadians) / 2), tanX = tanY * float(width) / float(height);
                 torch::Tensor dir = F.view({1, 1, 1, 3}) + (ndcx * tanX).unsqueeze(3) * R.view({1, 1, 1, 3}) + (ndcy * tanY).unsqueeze(3) * U.view({1, 1, 1, 3});
                 dir = dir / dir.norm(2, 3, true);
                 return std::make_tuple(E.view({1, 1, 1, 3}).expand({numSamples, height, width, 3}).contiguous(), dir.contiguous());
             }, py::arg("width"), py::arg("height"), py::arg("num_samples"), py::arg("time") = 0, py::arg("double_precision") = false);

    // ---- image evaluators (iimage_evaluator.cpp:325-358, image_evaluator_simple.cpp:427-475)
    py::class_<ImageEvaluatorSimple, std::shared_ptr<ImageEvaluatorSimple>> ie(m, "ImageEvaluatorSimple");
    py::enum_<ChannelMode>(ie, "ChannelMode")
        .value("Mask", ChannelMask).value("Normal", ChannelNormal).value("Depth", ChannelDepth).value("Color", ChannelColor)
        .export_values();
    ie.def(py::init<>())
        .def_readwrite("selected_channel", &ImageEvaluatorSimple::selectedChannel)
        .def_readwrite("double_precision", &ImageEvaluatorSimple::doublePrecision)
        .def("render", &ImageEvaluatorSimple::render, py::arg("width"), py::arg("height"))
        .def("render_stripes", &ImageEvaluatorSimple::renderStripes, py::arg("width"), py::arg("height"), py::arg("rank"), py::arg("world"), py::arg("stripe") = 16,
             "This rank's round-robin row stripes of the frame, compact: (B, 8, rows, W).  No reference counterpart (multi-GPU image tiles).")
        .def_static("stripe_rows", &ImageEvaluatorSimple::stripeRows, py::arg("height"), py::arg("stripe"), py::arg("rank"), py::arg("world"))
        .def_static("Assemble_stripes", &ImageEvaluatorSimple::assembleStripes, py::arg("gathered"), py::arg("height"), py::arg("stripe") = 16,
                    "(world, B, 8, rows, W) compact stripe images of all ranks -> (B, 8, H, W)")
        .def("refine", &ImageEvaluatorSimple::refine, py::arg("width"), py::arg("height"), py::arg("previous"))
        .def("get_module_for_tag", &ImageEvaluatorSimple::moduleForTag, py::arg("tag"))
        .def("compute_batch_count", [](ImageEvaluatorSimple& e) { return e.camera ? e.camera->batches() : 1; })
        .def("is_iterative_refining", [](ImageEvaluatorSimple&) { return false; })
        .def("get_supported_tags", [](ImageEvaluatorSimple&) { return std::vector<std::string>{"camera", "volume", "RayEvaluation"}; })
        .def_readwrite("camera", &ImageEvaluatorSimple::camera)
        .def_readwrite("ray_evaluator", &ImageEvaluatorSimple::rayEvaluator)
        .def_readwrite("volume", &ImageEvaluatorSimple::volume)
        .def_readwrite("spp_log2", &ImageEvaluatorSimple::sppLog2)
        .def_readwrite("use_tonemapping", &ImageEvaluatorSimple::useTonemapping)
        .def_readwrite("tonemapping_shoulder", &ImageEvaluatorSimple::tonemappingShoulder)
        .def_property_readonly("last_max_exposure", [](ImageEvaluatorSimple& e) { e.exposure(); return e.lastMaxExposure; })
        .def_readwrite("fix_max_exposure", &ImageEvaluatorSimple::fixMaxExposure)
        .def_readwrite("fixed_max_exposure", &ImageEvaluatorSimple::fixedMaxExposure)
        .def_static("Extract_color", &ImageEvaluatorSimple::extractColorStatic, py::arg("raw_input"), py::arg("use_tonemapping"),
                    py::arg("max_exposure"), py::arg("channel") = ChannelColor)
        .def("extract_color", [](ImageEvaluatorSimple& e, const torch::Tensor& raw) {
            return ImageEvaluatorSimple::extractColorStatic(raw, e.useTonemapping, e.useTonemapping ? e.exposure() : 1.f, e.selectedChannel);
        }, py::arg("raw_input"));
    m.attr("IImageEvaluator") = m.attr("ImageEvaluatorSimple");

    m.def("load_from_json", &loadFromJson, py::arg("filename"));
}
