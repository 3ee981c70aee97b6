// yaml_lite.h — the subset of YAML the reference's scene files use
// (sampleScenes/*.yaml): comments, block maps with any
// consistent indentation, block sequences of maps ("- key: value"), flow
// sequences of scalars ("[0, 2.5, 8.5]"), plain / quoted scalars.  yaml-cpp
// (the reference's parser, scene.cpp:136) is an empty submodule in the
// snapshot, so this is a from-scratch parser with the same observable
// behaviour on those files: node["key"], IsSequence(), as<int|float|string>,
// as<vector<float>>, and a line number for error messages (YAML::Mark).
#pragma once
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace yamllite {

struct ParseError : std::runtime_error {
    int line;
    ParseError(int l, const std::string& m) : std::runtime_error("yaml: error at line " + std::to_string(l) + ": " + m), line(l) {}
};

class Node {
public:
    enum Kind { Null, Scalar, Sequence, Map };
    Node() : kind(Null), line(0) {}
    Kind kind;
    int line;
    std::string scalar;
    std::vector<Node> seq;
    std::vector<std::pair<std::string, Node>> map;

    explicit operator bool() const { return kind != Null; }
    bool IsSequence() const { return kind == Sequence; }
    bool IsMap() const { return kind == Map; }
    bool IsScalar() const { return kind == Scalar; }
    // map lookup; returns a Null node when absent (yaml-cpp's operator[] on a const node)
    const Node& operator[](const std::string& key) const;
    size_t size() const { return kind == Sequence ? seq.size() : (kind == Map ? map.size() : 0); }

    // conversions throw ParseError like YAML::BadConversion
    int asInt() const;
    float asFloat() const;
    std::string asString() const;
    std::vector<float> asFloatVector() const;
    bool tryFloat(float& out) const;  // MatScalar's try { as<float>() } (scene.cpp:104-110)
};

Node Load(const std::string& text);
Node LoadFile(const std::string& path);  // throws ParseError(0, ...) if unreadable (YAML::BadFile)

}  // namespace yamllite
