// ia_driver -- the reference's test driver (test/test.cu:194-705) on this library.
//
//   ia_driver <config.json> [--per-image | --batch] [--batch-size N] [--gpus N] [--check-config]
//
// Reads the same JSON config (image_folder, input_image_filter, output_dir, extract_channel,
// saveFinalImages, saveIntermediateImages, calc_MSQE, knn, operations[...]), walks the folder
// in sorted order and runs the configured operations.
//
//   --batch      (default when the config allows it) loads the images, keeps them on the device
//                as one batch per image size and runs the whole operation list through the
//                batched executor (iak_pipeline_*): no per-operation allocation, copy or
//                synchronisation.
//   --gpus N     (batched flow) cuts the sorted file list into N contiguous shares, one host thread
//                and one device per share; feature rows, labels and the data-set histogram are
//                merged on the host in file order, so the CSV is the one-device CSV.
//                --share-devices lets N exceed the number of GPUs (shares then take devices in
//                turn): a rehearsal of the sharding on a smaller machine.
//   --per-image  the reference's own flow, call for call, through imageAnalysis.h: one image
//                at a time, one allocation per operation, previous image freed by the driver.
//                Chosen automatically when the config asks for what only that flow has:
//                intermediate images, the MSQE report.
//   --check-config  parse the config, list and decode the input files, write nothing, touch
//                no GPU.
//
// Output files are named as the reference names them (<output_dir>/<file> for final images,
// <output_dir>/<type>-<file> for intermediates, test/test.cu:123-134); the feature CSV has one
// row per image, seven features then the class label (test/test.cu:175-192).  Image files are
// uncompressed BMP (bmp_io.hpp).
#include <dirent.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <functional>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <regex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/iak.h"
#include "../../../include/imageAnalysis.h"
#include "../../../include/imageMachineLearning.h"
#include "bmp_io.hpp"
#include "mini_json.hpp"

using mini_json::Value;
using std::string;
using std::vector;

namespace {

#define DRV_CHECK(call)                                                                                    \
    do {                                                                                                   \
        int rc__ = (call);                                                                                 \
        if (rc__ != 0) {                                                                                   \
            fprintf(stderr, "%s failed: %s (%d) at %s:%d\n", #call, iak_error_string(rc__), rc__, __FILE__, \
                    __LINE__);                                                                             \
            exit(1);                                                                                       \
        }                                                                                                  \
    } while (0)

// test/test.cu:99-121
vector<string> file_names(const string &folder, const std::regex &filter) {
    vector<string> files;
    DIR *dir = opendir(folder.c_str());
    if (!dir) {
        perror(folder.c_str());
        return files;
    }
    while (dirent *ent = readdir(dir))
        if (strncmp(ent->d_name, ".", 1) && std::regex_search(ent->d_name, filter)) files.push_back(ent->d_name);
    closedir(dir);
    std::sort(files.begin(), files.end());
    return files;
}

// test/test.cu:175-192: features, then the label, comma separated, default ostream formatting
void write_csv(const string &path, const vector<vector<float>> &rows, const vector<int> &labels) {
    std::ofstream out(path);
    std::stringstream ss;
    for (size_t i = 0; i < rows.size(); i++) {
        for (float v : rows[i]) ss << v << ",";
        ss << labels[i] << std::endl;
    }
    out << ss.str();
}

struct Classes {  // test/test.cu:540-548
    std::map<string, int> ids;
    std::regex strip{"[0-9]+\\.BMP"};
    int label_of(const string &file) {
        const string name = std::regex_replace(file, strip, "");
        auto it = ids.find(name);
        if (it != ids.end()) return it->second;
        const int id = (int)ids.size();
        ids.emplace(name, id);
        return id;
    }
};

struct Config {
    string image_folder, output_dir, extract_channel, filter;
    bool save_final = false, save_intermediate = false, calc_msqe = false, knn = false;
    int knn_k = 0;
    string knn_input;
    Value operations;
    string features_csv;  // "output" of the first feature-extract operation
};

Config read_config(const string &path) {
    std::ifstream in(path, std::ifstream::binary);
    if (!in) {
        fprintf(stderr, "cannot open config %s\n", path.c_str());
        exit(-1);
    }
    std::stringstream ss;
    ss << in.rdbuf();
    Value json = mini_json::parse(ss.str());
    Config c;
    c.image_folder = json["image_folder"].asString();
    c.output_dir = json["output_dir"].asString();
    c.extract_channel = json["extract_channel"].asString();
    c.filter = json["input_image_filter"].asString();
    c.save_final = json["saveFinalImages"].asBool();
    c.save_intermediate = json["saveIntermediateImages"].asBool();
    c.calc_msqe = json["calc_MSQE"].asBool();
    c.knn = json["knn"].asBool();
    c.knn_k = json["knnConfig"]["k"].asInt();
    c.knn_input = json["knnConfig"]["input"].asString();
    c.operations = json["operations"];
    for (size_t i = 0; i < c.operations.size(); i++)
        if (c.operations[i]["type"].asString() == "feature-extract" && c.features_csv.empty())
            c.features_csv = c.operations[i]["output"].asString();
    return c;
}

// One configured operation with its host arrays decoded once.
struct Op {
    string type;
    int kW = 0, kH = 0, iparam = 0;
    float mean = -1, stddev = -1;
    vector<float> fk;
    vector<int> ik;
    bool known = true;
};

vector<Op> decode_operations(const Value &ops) {
    vector<Op> out;
    for (size_t i = 0; i < ops.size(); i++) {
        const Value &o = ops[i];
        Op op;
        op.type = o["type"].asString();
        auto kernel = [&](const Value &k, bool as_float) {
            op.kW = k["width"].asInt();
            op.kH = k["height"].asInt();
            const Value &vals = k["values"];
            if ((int)vals.size() != op.kW * op.kH) {  // the reference asserts (test/test.cu:79,92)
                fprintf(stderr, "operation %zu (%s): %zu kernel values for %dx%d\n", i, op.type.c_str(), vals.size(), op.kW,
                        op.kH);
                exit(1);
            }
            for (size_t j = 0; j < vals.size(); j++) {
                if (as_float) op.fk.push_back(vals[j].asFloat());
                else op.ik.push_back(vals[j].asInt());
            }
        };
        if (op.type == "linear-filter" || op.type == "average-filter") kernel(o["kernel"], true);
        else if (op.type == "median-filter") kernel(o["kernel"], false);
        else if (op.type == "dilation" || op.type == "erosion") kernel(o["structuring-element"], false);
        else if (op.type == "gaussian-noise") { op.mean = o["mean"].asFloat(); op.stddev = o["std_dev"].asFloat(); }
        else if (op.type == "salt-and-pepper") op.iparam = o["intensity"].asInt();
        else if (op.type == "threshold") op.iparam = o["threshold"].asInt();
        else if (op.type == "quantization") {
            const Value &lv = o["levels"];
            op.iparam = (int)lv.size();
            for (size_t j = 0; j < lv.size(); j++) {
                op.ik.push_back(lv[j]["min"].asInt());
                op.ik.push_back(lv[j]["max"].asInt());
                op.ik.push_back(lv[j]["val"].asInt());
            }
        } else if (op.type == "histogram-equalization" || op.type == "sobel-filter" || op.type == "compass-filter" ||
                   op.type == "otsu-thresh" || op.type == "kMeans-thresh" || op.type == "feature-extract") {
        } else {
            op.known = false;
        }
        out.push_back(op);
    }
    return out;
}

int executor_code(const string &type) {
    static const std::map<string, int> m = {
        {"histogram-equalization", IAK_OP_EQUALIZE}, {"gaussian-noise", IAK_OP_GAUSS},
        {"salt-and-pepper", IAK_OP_SALT_PEPPER},     {"linear-filter", IAK_OP_LINEAR},
        {"average-filter", IAK_OP_AVERAGE},          {"median-filter", IAK_OP_MEDIAN},
        {"sobel-filter", IAK_OP_SOBEL},              {"compass-filter", IAK_OP_COMPASS},
        {"dilation", IAK_OP_DILATE},                 {"erosion", IAK_OP_ERODE},
        {"threshold", IAK_OP_THRESHOLD},             {"otsu-thresh", IAK_OP_OTSU},
        {"quantization", IAK_OP_QUANTIZE},           {"kMeans-thresh", IAK_OP_KMEANS},
        {"feature-extract", IAK_OP_FEATURES}};
    auto it = m.find(type);
    return it == m.end() ? 0 : it->second;
}

// test/test.cu:136-173: rows of comma separated floats, shuffled with std::random_shuffle (libc
// rand()), flattened
vector<float> read_dataset_csv(int *numAttributes, int *numInstances, const string &path) {
    std::ifstream csv(path);
    string line;
    vector<vector<float>> rows;
    *numAttributes = 0;
    while (std::getline(csv, line)) {
        std::istringstream iss(line);
        string cell;
        vector<float> row;
        while (std::getline(iss, cell, ',')) row.push_back((float)std::stold(cell));
        if (rows.empty()) *numAttributes = (int)row.size();
        rows.push_back(row);
    }
    // std::random_shuffle as libstdc++ implements it: for i in [1, n): swap(v[i], v[rand() % (i + 1)])
    for (size_t i = 1; i < rows.size(); i++) std::swap(rows[i], rows[(size_t)rand() % (i + 1)]);
    *numInstances = (int)rows.size();
    vector<float> flat;
    for (auto &r : rows)
        for (int j = 0; j < *numAttributes; j++) flat.push_back(j < (int)r.size() ? r[j] : 0.0f);
    return flat;
}

// test/test.cu:587-597,633-636
void run_knn(const Config &c) {
    int numAttributes = 0, numInstances = 0;
    const auto t0 = std::chrono::steady_clock::now();
    vector<float> data = read_dataset_csv(&numAttributes, &numInstances, c.knn_input);
    float precision = 0, recall = 0, accuracy = 0;
    if (numInstances > 0 && numAttributes > 1)
        accuracy = knnTenfoldCrossVal(data.data(), numInstances, numAttributes, c.knn_k, &precision, &recall);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    printf("Total time required for a tenfold cross validation knn with k of %i on %i instances with %i features: %0.4f ms "
           "achieved accuracy of %0.4f%%, precision of: %f, and recall of %f\n",
           c.knn_k, numInstances, (numAttributes > 0) ? numAttributes - 1 : 0, ms, (float)(accuracy * 100.0), precision, recall);
}

struct Timers {
    std::map<string, double> ms;
    std::chrono::steady_clock::time_point t0;
    void start() {
        DRV_CHECK(iak_stream_sync(nullptr));
        t0 = std::chrono::steady_clock::now();
    }
    void stop(const string &key) {
        DRV_CHECK(iak_stream_sync(nullptr));
        ms[key] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    double get(const string &key) const {
        auto it = ms.find(key);
        return it == ms.end() ? 0.0 : it->second;
    }
};

// the reference's closing report (test/test.cu:605-636), same wording
void report(const Config &c, const Timers &t, double batch_ms, double n, double msqe_total) {
    auto line = [&](const char *what, const string &key) {
        printf("Total time spent %s: %0.4f ms average of: %0.4f ms per image\n", what, t.get(key), t.get(key) / n);
    };
    printf("\n\nTotal time spent on the entire batch: %0.4f ms average of %0.4f ms for each image\n", batch_ms, batch_ms / n);
    if (c.extract_channel == "grey") line("converting to grayscale", "grey");
    else line("converting to a single channel", "channel");
    line("performing histogram equalization", "histogram-equalization");
    line("adding gaussian noise", "gaussian-noise");
    line("adding salt and pepper noise", "salt-and-pepper");
    if (c.calc_msqe)
        printf("Total time spent performing image quantization: %0.4f ms average of: %0.4f ms per image with an average MSQE of %0.4f\n",
               t.get("quantization"), t.get("quantization") / n, msqe_total / n);
    else
        line("performing image quantization", "quantization");
    line("linear filtering images", "linear-filter");
    line("average filtering images", "average-filter");
    line("median filtering images", "median-filter");
    line("Sobel filtering images", "sobel-filter");
    line("Compass filtering images", "compass-filter");
    line("dilating images", "dilation");
    line("eroding images", "erosion");
    line("Basic thresholding images", "threshold");
    line("Otsu thresholding images", "otsu-thresh");
    line("k Means thresholding images", "kMeans-thresh");
    line("extracting features from images", "feature-extract");
}

void save_device_image(const Config &c, Image *d_image, const string &type, const string &file) {  // test/test.cu:123-134
    Image h;
    copyDeviceImageToHost(d_image, &h);
    string path = c.output_dir + "/" + (type.empty() ? file : type + "-" + file);
    const string err = bmp_io::write_gray(path, h.image, h.width, h.height);
    if (!err.empty()) fprintf(stderr, "%s: %s\n", path.c_str(), err.c_str());
    free(h.image);
}

// ---- the reference's flow, call for call (test/test.cu:266-564) ------------------------------------------
int run_per_image(const Config &c, const vector<string> &files, vector<Op> &ops) {
    Timers t;
    Classes classes;
    vector<vector<float>> dataset;
    vector<int> labels;
    int h_histogram[256];
    int h_mappings[256];
    double msqe_total = 0;
    double n_images = (double)files.size();
    bool randomness = false, edges = false;
    Image d_image{}, d_temp{};
    const auto batch0 = std::chrono::steady_clock::now();
    for (const string &file : files) {
        bmp_io::Planar img;
        const string err = bmp_io::read(c.image_folder + "/" + file, &img);
        if (!err.empty()) {
            fprintf(stderr, "%s: %s\n", file.c_str(), err.c_str());
            n_images--;
            continue;
        }
        RGBImage h_rgb{img.rgb.data(), 3, img.width, img.height};
        if (c.extract_channel == "grey") {
            t.start();
            RGBImage d_rgb{};
            copyHostRGBImageToDevice(&h_rgb, &d_rgb);
            convertRGBToGrayscale(&d_rgb, &d_image, 0);
            DRV_CHECK(iak_free(d_rgb.image));  // the reference leaks this buffer for every image
            t.stop("grey");
        } else {
            const int ch = c.extract_channel == "red" ? 0 : c.extract_channel == "green" ? 1 : c.extract_channel == "blue" ? 2 : -1;
            if (ch < 0) {
                printf("Unsupported color option: %s\n", c.extract_channel.c_str());
                exit(-10);
            }
            t.start();
            Image h_image{};
            extractSingleColorChannel(&h_rgb, &h_image, ch);
            copyHostImageToDevice(&h_image, &d_image);
            t.stop("channel");
        }
        if (c.save_intermediate) save_device_image(c, &d_image, c.extract_channel, file);
        for (Op &op : ops) {
            bool supported = true;
            auto adopt = [&]() {  // test/test.cu:338-339: free the input, keep the output
                CUDA_CHECK_RETURN(hipFree(d_image.image));
                d_image.image = d_temp.image;
            };
            if (op.type == "feature-extract") {
                t.start();
                dataset.push_back(featureExtraction(&d_image, &d_temp, h_histogram, nullptr));
                labels.push_back(classes.label_of(file));
                t.stop(op.type);
            } else if (!op.known) {
                printf("Unsupported Operation\n");
                supported = false;
            } else {
                if ((op.type == "gaussian-noise" || op.type == "salt-and-pepper") && !randomness) {
                    setupRandomness(&d_image);  // outside the operation's timer, as in the reference
                    randomness = true;
                }
                if (op.type == "sobel-filter" && !edges) {
                    setupEdgeDetection();
                    edges = true;
                }
                t.start();
                if (op.type == "linear-filter") linearFilter(&d_image, &d_temp, op.fk.data(), op.kW, op.kH);
                else if (op.type == "average-filter") averageFilter(&d_image, &d_temp, op.fk.data(), op.kW, op.kH);
                else if (op.type == "median-filter") medianFilter(&d_image, &d_temp, op.ik.data(), op.kW, op.kH);
                else if (op.type == "gaussian-noise") addGaussianNoise(&d_image, &d_temp, op.mean, op.stddev);
                else if (op.type == "salt-and-pepper") saltAndPepperNoise(&d_image, &d_temp, op.iparam);
                else if (op.type == "histogram-equalization") {
                    calculateHistogram(&d_image, h_histogram, nullptr);
                    equalizeHistogram(h_histogram, h_mappings, d_image.width * d_image.height);
                    equalizeImageWithHist(&d_image, &d_temp, h_mappings);
                } else if (op.type == "quantization") {
                    imageQuantization(&d_image, &d_temp, op.ik.data(), op.iparam);
                    if (c.calc_msqe) msqe_total += calcMSQE(&d_image, &d_temp);
                } else if (op.type == "sobel-filter") sobelFilter(&d_image, &d_temp);
                else if (op.type == "compass-filter") compassFilter(&d_image, &d_temp);
                else if (op.type == "dilation") imageDilation(&d_image, &d_temp, op.ik.data(), op.kW, op.kH);
                else if (op.type == "erosion") imageErosion(&d_image, &d_temp, op.ik.data(), op.kW, op.kH);
                else if (op.type == "threshold") thresholdImage(&d_image, &d_temp, op.iparam);
                else if (op.type == "otsu-thresh") otsuThresholdImage(&d_image, &d_temp);
                else if (op.type == "kMeans-thresh") kMeansThresholding(&d_image, &d_temp);
                adopt();
                t.stop(op.type);
            }
            if (c.save_intermediate && supported) save_device_image(c, &d_image, op.type, file);
        }
        if (c.save_final) save_device_image(c, &d_image, "", file);
        CUDA_CHECK_RETURN(hipFree(d_image.image));  // the reference leaks the last image of every file
        d_image.image = nullptr;
    }
    DRV_CHECK(iak_stream_sync(nullptr));
    const double batch_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - batch0).count();
    if (!c.features_csv.empty() && !dataset.empty()) write_csv(c.features_csv, dataset, labels);
    if (edges) cleanupEdgeDetection();
    if (randomness) cleanupRandomness();
    report(c, t, batch_ms, n_images > 0 ? n_images : 1, msqe_total);
    return 0;
}

// ---- the batched flow ---------------------------------------------------------------------------------------
// What one device makes of its contiguous share of the file list.
struct ShardResult {
    vector<string> files;            // one per feature row, in row order
    vector<vector<float>> rows;      // feature rows
    double n_images = 0, pipeline_ms = 0, load_ms = 0, save_ms = 0;
    unsigned long long hist[256];    // dataset histogram of the grayscale / channel images (value 0 not counted)
};

struct BatchPlan {
    vector<iak_op> ops;
    bool has_features = false;
    int n_kmeans = 0;
    int input = 0;
};

// files[lo, hi) on `device`: loads the images, keeps them on the device as one batch per image
// size and runs the whole operation list through the batched executor.
void run_shard(const Config &c, const vector<string> &files, size_t lo, size_t hi, const BatchPlan &plan, int batch_size,
               int device, ShardResult *res) {
    DRV_CHECK(iak_set_device(device));
    memset(res->hist, 0, sizeof res->hist);
    // One executor per image size, created when the first image of that size shows up; the k-means
    // centroid stream is the executor's, so a folder of equally sized images (the data set's case)
    // sees the same draws as the per-image flow -- a shard that starts at file `lo` first moves
    // that stream past the `lo` images before it.
    struct Lane {
        int W = 0, H = 0;
        iak_pipeline *pipe = nullptr;
        uint8_t *d_rgb = nullptr, *d_out = nullptr;
        float *d_feat = nullptr;
        unsigned long long *d_hist = nullptr;
        std::unique_ptr<uint8_t[]> h_rgb, h_out;  // not value-initialised: hundreds of megabytes
        vector<float> h_feat;
        vector<string> pending;
    };
    vector<Lane> lanes;
    auto flush = [&](Lane &L) {
        const int B = (int)L.pending.size();
        if (B == 0) return;
        const size_t N = (size_t)L.W * L.H;
        auto t0 = std::chrono::steady_clock::now();
        DRV_CHECK(iak_memcpy_h2d(L.d_rgb, L.h_rgb.get(), 3 * N * (size_t)B, nullptr));
        DRV_CHECK(iak_pipeline_run(L.pipe, L.d_rgb, L.d_out, B, L.d_hist, nullptr, nullptr));
        DRV_CHECK(iak_stream_sync(nullptr));
        res->pipeline_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        t0 = std::chrono::steady_clock::now();
        if (c.save_final) {
            DRV_CHECK(iak_memcpy_d2h(L.h_out.get(), L.d_out, N * (size_t)B, nullptr));
            for (int b = 0; b < B; b++) {
                const string path = c.output_dir + "/" + L.pending[b];
                const string err = bmp_io::write_gray(path, L.h_out.get() + N * (size_t)b, L.W, L.H);
                if (!err.empty()) fprintf(stderr, "%s: %s\n", path.c_str(), err.c_str());
            }
        }
        if (plan.has_features) {
            DRV_CHECK(iak_memcpy_d2h(L.h_feat.data(), L.d_feat, sizeof(float) * IAK_N_FEATURES * (size_t)B, nullptr));
            for (int b = 0; b < B; b++) {
                res->rows.emplace_back(L.h_feat.begin() + IAK_N_FEATURES * b, L.h_feat.begin() + IAK_N_FEATURES * (b + 1));
                res->files.push_back(L.pending[b]);
            }
        }
        res->save_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        res->n_images += B;
        L.pending.clear();
    };
    for (size_t f = lo; f < hi; f++) {
        const string &file = files[f];
        auto t0 = std::chrono::steady_clock::now();
        bmp_io::Planar img;
        const string err = bmp_io::read(c.image_folder + "/" + file, &img);
        if (!err.empty()) {
            fprintf(stderr, "%s: %s\n", file.c_str(), err.c_str());
            continue;
        }
        Lane *L = nullptr;
        for (Lane &l : lanes)
            if (l.W == img.width && l.H == img.height) L = &l;
        if (!L) {
            // images of another size wait for nothing: what is pending in other lanes keeps its order
            // within its own size class; rows of the CSV are written per flushed batch
            lanes.emplace_back();
            L = &lanes.back();
            L->W = img.width;
            L->H = img.height;
            const size_t N = (size_t)L->W * L->H;
            DRV_CHECK(iak_pipeline_create(&L->pipe, L->W, L->H, batch_size, plan.ops.data(), (int)plan.ops.size(), 1234ULL,
                                          IAK_PIPE_FUSED));
            DRV_CHECK(iak_pipeline_set_input(L->pipe, plan.input));
            DRV_CHECK(iak_pipeline_skip_images(L->pipe, (long long)lo, nullptr));
            DRV_CHECK(iak_malloc((void **)&L->d_rgb, 3 * N * (size_t)batch_size));
            DRV_CHECK(iak_malloc((void **)&L->d_out, N * (size_t)batch_size));
            DRV_CHECK(iak_malloc((void **)&L->d_hist, 256 * sizeof(unsigned long long)));
            DRV_CHECK(iak_memset(L->d_hist, 0, 256 * sizeof(unsigned long long), nullptr));
            L->h_rgb.reset(new uint8_t[3 * N * (size_t)batch_size]);
            L->h_out.reset(new uint8_t[N * (size_t)batch_size]);
            if (plan.has_features) {
                DRV_CHECK(iak_malloc((void **)&L->d_feat, sizeof(float) * IAK_N_FEATURES * (size_t)batch_size));
                DRV_CHECK(iak_pipeline_bind_features(L->pipe, L->d_feat));
                L->h_feat.resize((size_t)IAK_N_FEATURES * batch_size);
            }
        }
        const size_t N = (size_t)L->W * L->H;
        memcpy(L->h_rgb.get() + 3 * N * L->pending.size(), img.rgb.data(), 3 * N);
        L->pending.push_back(file);
        res->load_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if ((int)L->pending.size() == batch_size) flush(*L);
    }
    for (Lane &L : lanes) flush(L);
    for (Lane &L : lanes) {
        unsigned long long h[256];
        DRV_CHECK(iak_memcpy_d2h(h, L.d_hist, sizeof h, nullptr));
        for (int v = 0; v < 256; v++) res->hist[v] += h[v];
        DRV_CHECK(iak_pipeline_destroy(L.pipe));
        DRV_CHECK(iak_free(L.d_rgb));
        DRV_CHECK(iak_free(L.d_out));
        DRV_CHECK(iak_free(L.d_hist));
        if (L.d_feat) DRV_CHECK(iak_free(L.d_feat));
    }
}

// The folder is cut into `gpus` contiguous shares (the first n % gpus shares hold one file more),
// one host thread per device runs its share, and the host merges what the reference keeps at
// data-set level: the feature rows with their class labels in file order (test/test.cu:541-550)
// and the 256-bin histogram (test/test.cu:407-411).  The payload is a few KiB per device, so
// the merge is a host loop, not a collective.  *images_done: images that went through.
// --dataset-summary: after the reference's report lines also print the data-set level sums (the histogram the
// reference keeps in totalHist, compiled out, and the per-class feature statistics).  Off by default: the
// report then has exactly the reference driver's lines.
bool g_dataset_summary = false;

int run_batched(const Config &c, const vector<string> &files, vector<Op> &ops, int batch_size, int gpus, bool share_devices,
                double *images_done) {
    // the executor takes every operation of the list or none: an unknown type is reported and
    // skipped, as the reference's loop does
    BatchPlan plan;
    for (Op &op : ops) {
        const int code = executor_code(op.type);
        if (!code) {
            printf("Unsupported Operation\n");
            continue;
        }
        iak_op o{};
        o.type = code;
        o.kW = op.kW;
        o.kH = op.kH;
        o.fkernel = op.fk.empty() ? nullptr : op.fk.data();
        o.ikernel = op.ik.empty() ? nullptr : op.ik.data();
        o.iparam = op.iparam;
        o.mean = op.mean;
        o.stddev = op.stddev;
        plan.ops.push_back(o);
        plan.has_features = plan.has_features || code == IAK_OP_FEATURES;
        plan.n_kmeans += code == IAK_OP_KMEANS ? 1 : 0;
    }
    plan.input = c.extract_channel == "grey" ? IAK_INPUT_LUMINANCE : c.extract_channel == "red" ? IAK_INPUT_RED
                 : c.extract_channel == "green" ? IAK_INPUT_GREEN : c.extract_channel == "blue" ? IAK_INPUT_BLUE : -1;
    if (plan.input < 0) {  // test/test.cu:319-322
        printf("Unsupported color option: %s\n", c.extract_channel.c_str());
        exit(-10);
    }
    int n_devices = 0;
    DRV_CHECK(iak_device_count(&n_devices));
    if (gpus > n_devices && !share_devices) {
        fprintf(stderr, "--gpus %d: this machine has %d GPU(s)\n", gpus, n_devices);
        return 2;
    }
    const auto t_all = std::chrono::steady_clock::now();
    vector<ShardResult> shard((size_t)gpus);
    vector<std::thread> workers;
    const size_t n = files.size(), base = n / (size_t)gpus, extra = n % (size_t)gpus;
    for (int g = 0; g < gpus; g++) {
        const size_t lo = (size_t)g * base + std::min((size_t)g, extra);
        const size_t hi = lo + base + ((size_t)g < extra ? 1 : 0);
        const int per = std::max<size_t>(1, std::min<size_t>((size_t)batch_size, hi - lo));
        workers.emplace_back(run_shard, std::cref(c), std::cref(files), lo, hi, std::cref(plan), per, g % n_devices, &shard[(size_t)g]);
    }
    for (auto &w : workers) w.join();
    // merge, shard by shard = in file order
    Classes classes;
    vector<vector<float>> dataset;
    vector<int> labels;
    unsigned long long hist[256] = {0};
    double n_images = 0, pipeline_ms = 0, load_ms = 0, save_ms = 0;
    for (const ShardResult &r : shard) {
        for (size_t k = 0; k < r.rows.size(); k++) {
            dataset.push_back(r.rows[k]);
            labels.push_back(classes.label_of(r.files[k]));
        }
        for (int v = 0; v < 256; v++) hist[v] += r.hist[v];
        n_images += r.n_images;
        pipeline_ms = std::max(pipeline_ms, r.pipeline_ms);  // the devices run side by side
        load_ms = std::max(load_ms, r.load_ms);
        save_ms = std::max(save_ms, r.save_ms);
    }
    if (!c.features_csv.empty() && !dataset.empty()) write_csv(c.features_csv, dataset, labels);
    const double all_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_all).count();
    const double nn = n_images > 0 ? n_images : 1;
    printf("\n\nTotal time spent on the entire batch: %0.4f ms average of %0.4f ms for each image\n", all_ms, all_ms / nn);
    printf("Total time spent reading and decoding images: %0.4f ms average of: %0.4f ms per image\n", load_ms, load_ms / nn);
    printf("Total time spent copying to the device and running all operations: %0.4f ms average of: %0.4f ms per image\n",
           pipeline_ms, pipeline_ms / nn);
    printf("Total time spent copying back, encoding and writing results: %0.4f ms average of: %0.4f ms per image\n", save_ms,
           save_ms / nn);
    // data-set level sums (the reference keeps totalHist, compiled out, and the per-class rows)
    if (!g_dataset_summary) {
        *images_done = n_images;
        for (long long k = 0; k < 2LL * plan.n_kmeans * (long long)n_images; k++) (void)rand();  // see the end of this function
        return 0;
    }
    unsigned long long counted = 0, weighted = 0;
    for (int v = 0; v < 256; v++) {
        counted += hist[v];
        weighted += hist[v] * (unsigned long long)v;
    }
    printf("Dataset histogram over %d device(s): %llu non-zero pixels in %.0f images, mean of those %0.4f\n", gpus, counted,
           n_images, counted ? (double)weighted / (double)counted : 0.0);
    if (plan.has_features) {
        vector<string> names(classes.ids.size());
        for (auto &kv : classes.ids) names[(size_t)kv.second] = kv.first;
        for (size_t k = 0; k < names.size(); k++) {
            double cnt = 0, s[IAK_N_FEATURES] = {0}, ss[IAK_N_FEATURES] = {0};
            for (size_t r = 0; r < dataset.size(); r++)
                if (labels[r] == (int)k) {
                    cnt += 1;
                    for (int f = 0; f < IAK_N_FEATURES; f++) {
                        s[f] += dataset[r][f];
                        ss[f] += (double)dataset[r][f] * dataset[r][f];
                    }
                }
            printf("class %zu '%s': %.0f images; mean / std of", k, names[k].c_str(), cnt);
            static const char *fname[IAK_N_FEATURES] = {"mean", "stddev", "area", "perimeter", "mode", "rangeMin", "rangeMax"};
            for (int f = 0; f < IAK_N_FEATURES; f++) {
                const double m = cnt ? s[f] / cnt : 0, var = cnt ? std::max(0.0, ss[f] / cnt - m * m) : 0;
                printf(" %s %0.3f / %0.3f%s", fname[f], m, sqrt(var), f + 1 < IAK_N_FEATURES ? "," : "\n");
            }
        }
    }
    *images_done = n_images;
    // The per-image flow has drawn two rand() values per image per k-means operation by now
    // (imageAnalysis.cu:359-361); here the executors drew them on the device.  Bring the host
    // stream to the same point, so that what follows (the shuffle of the cross-validation data,
    // test/test.cu:165) sees the same values in both flows.
    for (long long k = 0; k < 2LL * plan.n_kmeans * (long long)n_images; k++) (void)rand();
    return 0;
}

}  // namespace

int main(int argc, char *argv[]) {
    if (argc < 2) {
        printf("please pass a config file path as the argument\n");
        exit(-1);
    }
    string config_path;
    bool force_per_image = false, force_batch = false, check_only = false, share_devices = false;
    int batch_size = 512, gpus = 1;
    for (int i = 1; i < argc; i++) {
        const string a = argv[i];
        if (a == "--per-image") force_per_image = true;
        else if (a == "--batch") force_batch = true;
        else if (a == "--check-config") check_only = true;
        else if (a == "--batch-size" && i + 1 < argc) batch_size = atoi(argv[++i]);
        else if (a == "--gpus" && i + 1 < argc) gpus = atoi(argv[++i]);
        else if (a == "--share-devices") share_devices = true;
        else if (a == "--dataset-summary") g_dataset_summary = true;
        else if (config_path.empty()) config_path = a;
        else {
            fprintf(stderr, "unexpected argument %s\n", a.c_str());
            return 2;
        }
    }
    if (batch_size < 1) batch_size = 1;
    if (gpus < 1) gpus = 1;
    Config c;
    try {
        c = read_config(config_path);
    } catch (const std::exception &e) {
        fprintf(stderr, "%s: %s\n", config_path.c_str(), e.what());
        return 2;
    }
    printf("Input: %s\nOutput: %s\nSaving intermediates: %s\nSaving Finals: %s\n", c.image_folder.c_str(), c.output_dir.c_str(),
           c.save_intermediate ? "true" : "false", c.save_final ? "true" : "false");
    vector<string> files = file_names(c.image_folder, std::regex(c.filter));
    vector<Op> ops = decode_operations(c.operations);
    if (check_only) {
        size_t ok = 0;
        for (const string &f : files) {
            bmp_io::Planar img;
            const string err = bmp_io::read(c.image_folder + "/" + f, &img);
            if (err.empty()) ok++;
            printf("%s: %s\n", f.c_str(), err.empty() ? (std::to_string(img.width) + "x" + std::to_string(img.height)).c_str() : err.c_str());
        }
        for (const Op &op : ops) printf("operation %s%s\n", op.type.c_str(), op.known ? "" : " (unsupported)");
        printf("%zu of %zu files readable, %zu operations\n", ok, files.size(), ops.size());
        return 0;
    }
    const bool needs_per_image = c.save_intermediate || c.calc_msqe;
    if (force_batch && needs_per_image) {
        fprintf(stderr, "--batch: the config asks for intermediate images or the MSQE report, which only the per-image "
                        "flow produces\n");
        return 2;
    }
    {
        // The HIP runtime draws from libc rand() while it brings up its first queue.  The k-means
        // of the per-image flow seeds its centroids from rand() and expects the stream of a program
        // that has neither seeded nor drawn (imageAnalysis.cu:39-41,359-361): bring the runtime up
        // first, then put rand() back into that state.
        void *warm = nullptr;
        DRV_CHECK(iak_malloc(&warm, 256));
        DRV_CHECK(iak_memset(warm, 0, 256, nullptr));
        DRV_CHECK(iak_stream_sync(nullptr));
        DRV_CHECK(iak_free(warm));
        srand(1);
    }
    if ((size_t)batch_size > files.size()) batch_size = files.empty() ? 1 : (int)files.size();
    if (gpus > 1 && (force_per_image || needs_per_image)) {
        fprintf(stderr, "--gpus: only the batched flow shards the folder over devices\n");
        return 2;
    }
    double images_done = 0;
    const int rc = (force_per_image || needs_per_image) ? run_per_image(c, files, ops)
                                                        : run_batched(c, files, ops, batch_size, gpus, share_devices, &images_done);
    if (rc == 0 && c.knn) run_knn(c);  // after the batch, as the reference does (test/test.cu:587)
    return rc;
}
