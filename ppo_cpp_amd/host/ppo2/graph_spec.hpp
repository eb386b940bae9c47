// graph_spec.hpp -- importer for the reference's graph files (resources/ppo_cl/graphs/*.meta.txt, a TensorFlow
// MetaGraphDef in protobuf TEXT format; loaded by the reference through ReadTextProto, ppo2/session_creator.hpp:40).
// The HIP path does not execute graphs; it needs what the external generator baked into one: network shape, the loss /
// clipping / Adam constants and the initial weights (the `init` op assigns every variable from a Const, G:32233).
// A small brace-matching reader of the text format is enough -- no protobuf schema, no TensorFlow.
#pragma once
#include <cstdint>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/ppo_hip.h"

namespace graphspec {

struct Node {                               // one `key { ... }` block: scalar fields and child blocks, in file order
    std::vector<std::pair<std::string, std::string>> fields;
    std::vector<std::pair<std::string, std::unique_ptr<Node>>> blocks;
    const std::string* field(const std::string& k) const { for (auto& f : fields) if (f.first == k) return &f.second; return nullptr; }
    const Node* block(const std::string& k) const { for (auto& b : blocks) if (b.first == k) return b.second.get(); return nullptr; }
};

inline std::string trim(const std::string& s) {
    size_t a = s.find_first_not_of(" \t\r"), b = s.find_last_not_of(" \t\r");
    return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}

inline std::unique_ptr<Node> parse_block(std::istream& in) {
    std::unique_ptr<Node> n(new Node());
    std::string line;
    while (std::getline(in, line)) {
        const std::string l = trim(line);
        if (l.empty()) continue;
        if (l == "}") return n;
        if (l.back() == '{') n->blocks.push_back({trim(l.substr(0, l.size() - 1)), parse_block(in)});
        else { const size_t c = l.find(':'); if (c != std::string::npos) n->fields.push_back({trim(l.substr(0, c)), trim(l.substr(c + 1))}); }
    }
    return n;
}

// C-escaped protobuf string literal ("...") -> bytes
inline std::string unescape(const std::string& q) {
    std::string out;
    if (q.size() < 2 || q.front() != '"') return q;
    for (size_t i = 1; i + 1 < q.size(); ++i) {
        char c = q[i];
        if (c != '\\') { out.push_back(c); continue; }
        c = q[++i];
        switch (c) {
            case 'n': out.push_back('\n'); break; case 't': out.push_back('\t'); break; case 'r': out.push_back('\r'); break;
            case '\\': out.push_back('\\'); break; case '"': out.push_back('"'); break; case '\'': out.push_back('\''); break;
            default:
                if (c >= '0' && c <= '7') { int v = 0, k = 0; while (k < 3 && i < q.size() - 1 && q[i] >= '0' && q[i] <= '7') { v = v * 8 + (q[i] - '0'); ++i; ++k; } --i; out.push_back((char)v); }
                else out.push_back(c);
        }
    }
    return out;
}

struct ConstTensor { std::vector<int64_t> shape; std::vector<float> data; };

struct GraphSpec {
    ppo_config config;                                      // shape + graph-baked constants
    std::map<std::string, ConstTensor> initial;             // "pi_fc0/w" ... incl. q/w, q/b
    float beta1_power0 = 0.f, beta2_power0 = 0.f;
};

inline bool const_value(const Node& node, ConstTensor& out) {
    for (auto& b : node.blocks) {
        if (b.first != "attr") continue;
        const std::string* key = b.second->field("key");
        if (!key || unescape(*key) != "value") continue;
        const Node* v = b.second->block("value"); if (!v) return false;
        const Node* t = v->block("tensor"); if (!t) return false;
        const std::string* dt = t->field("dtype");
        if (!dt || *dt != "DT_FLOAT") return false;
        out.shape.clear(); out.data.clear();
        if (const Node* sh = t->block("tensor_shape"))
            for (auto& d : sh->blocks) if (d.first == "dim") { const std::string* sz = d.second->field("size"); out.shape.push_back(sz ? atoll(sz->c_str()) : 0); }
        size_t count = 1; for (int64_t d : out.shape) count *= (size_t)d;
        if (const std::string* tc = t->field("tensor_content")) {
            const std::string raw = unescape(*tc);
            out.data.resize(raw.size() / 4);
            std::memcpy(out.data.data(), raw.data(), out.data.size() * 4);
        } else {
            for (auto& f : t->fields) if (f.first == "float_val") out.data.push_back((float)atof(f.second.c_str()));
            if (out.data.size() == 1 && count > 1) out.data.assign(count, out.data[0]);
        }
        return !out.data.empty();
    }
    return false;
}

inline GraphSpec load_graph_spec(const std::string& path) {
    std::ifstream in(path);
    if (!in) throw std::runtime_error("graph_spec: cannot open " + path);
    std::unique_ptr<Node> top = parse_block(in);
    const Node* gd = top->block("graph_def");
    if (!gd) throw std::runtime_error("graph_spec: no graph_def block");
    std::map<std::string, const Node*> nodes;
    for (auto& b : gd->blocks) if (b.first == "node") { const std::string* nm = b.second->field("name"); if (nm) nodes[unescape(*nm)] = b.second.get(); }
    auto scalar = [&](const std::string& name) {
        auto it = nodes.find(name); ConstTensor t;
        if (it == nodes.end() || !const_value(*it->second, t)) throw std::runtime_error("graph_spec: constant not found: " + name);
        return t.data[0];
    };
    GraphSpec g{};
    // variables "model/<v>" are initialised from "model/<v>/Initializer/..." Const nodes
    for (auto& kv : nodes) {
        const std::string& name = kv.first;
        if (name.compare(0, 6, "model/") != 0) continue;
        const size_t ipos = name.find("/Initializer/");
        if (ipos == std::string::npos) continue;
        const std::string* op = kv.second->field("op");
        if (!op || unescape(*op) != "Const") continue;
        ConstTensor t;
        if (!const_value(*kv.second, t)) continue;
        const std::string var = name.substr(6, ipos - 6);
        if (var.find("Adam") != std::string::npos) continue;
        auto& slot = g.initial[var];
        if (t.data.size() >= slot.data.size()) slot = t;      // zeros initialisers: the filled Const wins over the scalar
    }
    // zero initialisers are stored as {shape Const, scalar}: recover the shape from the variable node
    for (auto& kv : g.initial) {
        auto it = nodes.find("model/" + kv.first);
        if (it == nodes.end()) continue;
        for (auto& b : it->second->blocks) {
            if (b.first != "attr") continue;
            const std::string* key = b.second->field("key");
            if (!key || unescape(*key) != "shape") continue;
            const Node* sh = b.second->block("value") ? b.second->block("value")->block("shape") : nullptr;
            if (!sh) continue;
            std::vector<int64_t> dims;
            for (auto& d : sh->blocks) if (d.first == "dim") { const std::string* sz = d.second->field("size"); dims.push_back(sz ? atoll(sz->c_str()) : 0); }
            size_t count = 1; for (int64_t d : dims) count *= (size_t)d;
            if (kv.second.data.size() == 1 && count > 1) kv.second.data.assign(count, kv.second.data[0]);
            if (kv.second.data.size() == count) kv.second.shape = dims;
        }
    }
    std::vector<int32_t> hidden;
    for (int l = 0; l < PPO_MAX_LAYERS; ++l) {
        auto it = g.initial.find("pi_fc" + std::to_string(l) + "/w");
        if (it == g.initial.end()) break;
        hidden.push_back((int32_t)it->second.shape.at(1));
    }
    if (hidden.empty() || !g.initial.count("pi/w")) throw std::runtime_error("graph_spec: no pi_fc*/w or pi/w variables");
    ppo_config_default(&g.config, (int32_t)g.initial["pi_fc0/w"].shape.at(0), (int32_t)g.initial["pi/w"].shape.at(1), (int32_t)hidden.size(), hidden.data());
    g.config.ent_coef = scalar("loss/mul_4/y");                               // G:11323
    g.config.vf_coef = scalar("loss/mul_5/y");                                // G:11395
    g.config.max_grad_norm = scalar("loss/clip_by_global_norm/mul/x");        // G:24370
    g.config.adam_beta1 = scalar("ppo2/_train/beta1");                        // G:30430-30490
    g.config.adam_beta2 = scalar("ppo2/_train/beta2");
    g.config.adam_eps = scalar("ppo2/_train/epsilon");
    g.beta1_power0 = scalar("beta1_power/initial_value");                     // G:25426
    g.beta2_power0 = scalar("beta2_power/initial_value");                     // G:25579
    return g;
}

// create a handle configured from a graph file and assign the graph's initial weights (= load_graph + Run("init"))
inline ppo_handle* create_from_graph(const GraphSpec& g, int device = -1) {
    ppo_config cfg = g.config; cfg.device = device;
    ppo_handle* h = nullptr;
    if (ppo_create(&cfg, &h) != 0) throw std::runtime_error(ppo_last_error(nullptr));
    const int nt = ppo_num_tensors(h);
    for (int i = 0; i < nt; ++i) {
        char name[32]; int32_t r, c;
        ppo_tensor_info(h, i, name, &r, &c);
        auto it = g.initial.find(name);
        if (it == g.initial.end()) { ppo_destroy(h); throw std::runtime_error(std::string("graph_spec: graph lacks variable ") + name); }
        if (ppo_set_tensor(h, 0, i, it->second.data.data(), (int64_t)it->second.data.size()) != 0) { const std::string e = ppo_last_error(h); ppo_destroy(h); throw std::runtime_error(e); }
    }
    const float pw[2] = {g.beta1_power0, g.beta2_power0};
    ppo_set_beta_powers(h, pw);
    return h;
}

}  // namespace graphspec
