#include "yaml_lite.h"

#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <sstream>

namespace yamllite {

namespace {

struct Line { int indent; std::string text; int no; };

std::string rstrip(const std::string& s) {
    size_t e = s.size();
    while (e > 0 && (s[e - 1] == ' ' || s[e - 1] == '\t' || s[e - 1] == '\r')) --e;
    return s.substr(0, e);
}
std::string strip(const std::string& s) {
    size_t b = 0;
    while (b < s.size() && (s[b] == ' ' || s[b] == '\t')) ++b;
    return rstrip(s.substr(b));
}

// removes a trailing comment: '#' at column 0 or preceded by whitespace, outside quotes
std::string uncomment(const std::string& s) {
    char q = 0;
    for (size_t i = 0; i < s.size(); ++i) {
        char c = s[i];
        if (q) { if (c == q) q = 0; continue; }
        if (c == '"' || c == '\'') { q = c; continue; }
        if (c == '#' && (i == 0 || s[i - 1] == ' ' || s[i - 1] == '\t')) return s.substr(0, i);
    }
    return s;
}

std::vector<Line> split_lines(const std::string& text) {
    std::vector<Line> out;
    std::istringstream in(text);
    std::string raw;
    int no = 0;
    while (std::getline(in, raw)) {
        ++no;
        if (no == 1 && raw.size() >= 3 && (unsigned char)raw[0] == 0xEF && (unsigned char)raw[1] == 0xBB && (unsigned char)raw[2] == 0xBF) raw = raw.substr(3);
        std::string s = rstrip(uncomment(raw));
        size_t ind = 0;
        while (ind < s.size() && s[ind] == ' ') ++ind;
        if (ind < s.size() && s[ind] == '\t') throw ParseError(no, "tab used for indentation");
        if (ind == s.size()) continue;  // blank / comment only
        std::string t = s.substr(ind);
        if (t == "---" || t == "...") continue;
        out.push_back({(int)ind, t, no});
    }
    return out;
}

std::string unquote(const std::string& s) {
    if (s.size() >= 2 && ((s.front() == '"' && s.back() == '"') || (s.front() == '\'' && s.back() == '\''))) return s.substr(1, s.size() - 2);
    return s;
}

// position of the ':' that separates key and value, or npos
size_t key_colon(const std::string& s) {
    char q = 0; int depth = 0;
    for (size_t i = 0; i < s.size(); ++i) {
        char c = s[i];
        if (q) { if (c == q) q = 0; continue; }
        if (c == '"' || c == '\'') { q = c; continue; }
        if (c == '[' || c == '{') ++depth;
        else if (c == ']' || c == '}') --depth;
        else if (c == ':' && depth == 0 && (i + 1 == s.size() || s[i + 1] == ' ')) return i;
    }
    return std::string::npos;
}

Node parse_inline(const std::string& v, int line) {
    Node n; n.line = line;
    std::string s = strip(v);
    if (s.empty() || s == "~" || s == "null") { n.kind = Node::Null; return n; }
    if (s.front() == '[') {
        if (s.back() != ']') throw ParseError(line, "unterminated flow sequence");
        n.kind = Node::Sequence;
        std::string body = s.substr(1, s.size() - 2);
        std::string cur; int depth = 0; char q = 0;
        auto push = [&]() { std::string t = strip(cur); if (!t.empty()) n.seq.push_back(parse_inline(t, line)); cur.clear(); };
        for (char c : body) {
            if (q) { cur += c; if (c == q) q = 0; continue; }
            if (c == '"' || c == '\'') { q = c; cur += c; continue; }
            if (c == '[') ++depth;
            if (c == ']') --depth;
            if (c == ',' && depth == 0) { push(); continue; }
            cur += c;
        }
        push();
        return n;
    }
    if (s.front() == '{') throw ParseError(line, "flow maps are not supported");
    n.kind = Node::Scalar;
    n.scalar = unquote(s);
    return n;
}

struct Parser {
    std::vector<Line> lines;
    size_t pos = 0;

    Node parseBlock(int indent) {
        if (pos >= lines.size()) return Node();
        const Line& l = lines[pos];
        if (l.text[0] == '-' && (l.text.size() == 1 || l.text[1] == ' ')) return parseSeq(indent);
        return parseMap(indent);
    }

    Node parseMap(int indent) {
        Node n; n.kind = Node::Map; n.line = lines[pos].no;
        while (pos < lines.size() && lines[pos].indent == indent) {
            const Line l = lines[pos];
            if (l.text[0] == '-' && (l.text.size() == 1 || l.text[1] == ' ')) break;
            size_t c = key_colon(l.text);
            if (c == std::string::npos) throw ParseError(l.no, "expected 'key: value'");
            std::string key = unquote(strip(l.text.substr(0, c)));
            std::string val = strip(l.text.substr(c + 1));
            ++pos;
            Node child;
            if (!val.empty()) {
                child = parse_inline(val, l.no);
            } else if (pos < lines.size() && lines[pos].indent > indent) {
                child = parseBlock(lines[pos].indent);
            } else if (pos < lines.size() && lines[pos].indent == indent && lines[pos].text[0] == '-' &&
                       (lines[pos].text.size() == 1 || lines[pos].text[1] == ' ')) {
                child = parseSeq(indent);
            }
            child.line = child.line ? child.line : l.no;
            n.map.emplace_back(key, std::move(child));
        }
        if (pos < lines.size() && lines[pos].indent > indent) throw ParseError(lines[pos].no, "unexpected indentation");
        return n;
    }

    Node parseSeq(int indent) {
        Node n; n.kind = Node::Sequence; n.line = lines[pos].no;
        while (pos < lines.size() && lines[pos].indent == indent && lines[pos].text[0] == '-' &&
               (lines[pos].text.size() == 1 || lines[pos].text[1] == ' ')) {
            Line& l = lines[pos];
            size_t off = 1;
            while (off < l.text.size() && l.text[off] == ' ') ++off;
            std::string rest = l.text.substr(off);
            if (rest.empty()) {
                ++pos;
                if (pos < lines.size() && lines[pos].indent > indent) n.seq.push_back(parseBlock(lines[pos].indent));
                else n.seq.push_back(Node());
            } else if (rest[0] != '[' && key_colon(rest) != std::string::npos) {
                // "- key: value": a map whose first entry sits on the dash line
                l.indent = indent + (int)off;
                l.text = rest;
                n.seq.push_back(parseMap(l.indent));
            } else {
                n.seq.push_back(parse_inline(rest, l.no));
                ++pos;
            }
        }
        return n;
    }
};

}  // namespace

const Node& Node::operator[](const std::string& key) const {
    static const Node null_node;
    if (kind != Map) return null_node;
    for (const auto& kv : map)
        if (kv.first == key) return kv.second;
    return null_node;
}

int Node::asInt() const {
    if (kind != Scalar) throw ParseError(line, "bad conversion (expected an integer)");
    errno = 0;
    char* end = nullptr;
    long v = std::strtol(scalar.c_str(), &end, 0);
    if (end == scalar.c_str() || *end != 0 || errno) throw ParseError(line, "bad conversion: '" + scalar + "' is not an integer");
    return (int)v;
}
bool Node::tryFloat(float& out) const {
    if (kind != Scalar || scalar.empty()) return false;
    const char* s = scalar.c_str();
    // reject things strtof accepts but yaml-cpp does not treat as numbers here (hex floats, "nan", "inf" words)
    for (const char* p = s; *p; ++p) {
        char c = *p;
        if (!((c >= '0' && c <= '9') || c == '+' || c == '-' || c == '.' || c == 'e' || c == 'E')) return false;
    }
    errno = 0;
    char* end = nullptr;
    float v = std::strtof(s, &end);
    if (end == s || *end != 0) return false;
    out = v;
    return true;
}
float Node::asFloat() const {
    float v;
    if (!tryFloat(v)) throw ParseError(line, "bad conversion: '" + (kind == Scalar ? scalar : std::string("<non-scalar>")) + "' is not a number");
    return v;
}
std::string Node::asString() const {
    if (kind != Scalar) throw ParseError(line, "bad conversion (expected a scalar)");
    return scalar;
}
std::vector<float> Node::asFloatVector() const {
    if (kind != Sequence) throw ParseError(line, "bad conversion (expected a sequence)");
    std::vector<float> v;
    for (const Node& e : seq) v.push_back(e.asFloat());
    return v;
}

Node Load(const std::string& text) {
    Parser p;
    p.lines = split_lines(text);
    if (p.lines.empty()) return Node();
    Node root = p.parseBlock(p.lines[0].indent);
    if (p.pos < p.lines.size()) throw ParseError(p.lines[p.pos].no, "unexpected content (bad indentation?)");
    return root;
}

Node LoadFile(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw ParseError(0, "bad file: " + path);
    std::stringstream ss;
    ss << f.rdbuf();
    return Load(ss.str());
}

}  // namespace yamllite
