// Scene description language: tokenizer (sdl/src/tokenize.rs), parser (ast.rs) and interpreter (interpret.rs)
// of the reference, restated over the host Scene mirror.  See sdl.hpp for what is kept and what differs.
#include "sdl.hpp"

#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <functional>
#include <map>
#include <memory>
#include <random>
#include <sstream>

namespace rth {
namespace {

struct SdlError { std::string msg; };
[[noreturn]] void fail(const std::string& m) { throw SdlError{m}; }

// ------------------------------------------------------------------ tokens (tokenize.rs:20-82)
enum class Op { Add, Sub, Mul, Div, Mod, Assign, Range, Push, Eq, Neq, Lt, Gt, LtEq, GtEq, Not, And, Or };
enum class Sep { Comma, Colon, BraceOpen, BraceClose, ParensOpen, ParensClose, BracketOpen, BracketClose };
struct Token {
    enum Kind { KSep, KOp, KIdent, KString, KNumber, KBool } kind;
    Op op = Op::Add; Sep sep = Sep::Comma; std::string text; double num = 0.; bool b = false;
    bool is(Sep s) const { return kind == KSep && sep == s; }
    bool is(Op o) const { return kind == KOp && op == o; }
    bool is_ident(const char* s) const { return kind == KIdent && text == s; }
};
std::string show(const Token& t) {   // tokenize.rs:84-123
    static const char* seps[] = {",", ":", "{", "}", "(", ")", "[", "]"};
    static const char* ops[] = {"+", "-", "*", "/", "%", "=", "..", "<<", "==", "!=", "<", ">", "<=", ">=", "!", "&&", "||"};
    switch (t.kind) {
        case Token::KSep: return seps[(int)t.sep];
        case Token::KOp: return ops[(int)t.op];
        case Token::KIdent: return t.text;
        case Token::KString: return "\"" + t.text + "\"";
        case Token::KNumber: { std::ostringstream o; o << t.num; return o.str(); }
        default: return t.b ? "true" : "false";
    }
}

bool rust_whitespace(unsigned char c) { return c == ' ' || (c >= 9 && c <= 13) || c == 0x85 || c == 0xA0; }
bool rust_alnum(unsigned char c) { return (c >= '0' && c <= '9') || (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z'); }

// tokenize.rs:149-320
std::vector<Token> tokenize(const std::string& src) {
    std::vector<Token> out;
    size_t i = 0;
    const size_t n = src.size();
    auto tok_op = [&](Op o) { Token t; t.kind = Token::KOp; t.op = o; out.push_back(t); };
    auto tok_sep = [&](Sep s) { Token t; t.kind = Token::KSep; t.sep = s; out.push_back(t); };
    auto read_number = [&](bool negative) {   // tokenize.rs:354-403
        std::string pre, post;
        bool dec = false;
        while (i < n) {
            const char c = src[i];
            if (c == '.') {
                if (dec) fail("tokenizer error: unexpected character .");
                dec = true; i++;
            } else if (c >= '0' && c <= '9') { (dec ? post : pre).push_back(c); i++; }
            else break;
        }
        if (pre.empty() && post.empty()) fail("tokenizer error: error parsing number");
        if (pre.empty()) pre = "0";
        if (post.empty()) post = "0";
        const double v = std::strtod((pre + "." + post).c_str(), nullptr);
        Token t; t.kind = Token::KNumber; t.num = negative ? -v : v;
        out.push_back(t);
    };
    while (i < n) {
        const unsigned char c = (unsigned char)src[i];
        if (rust_whitespace(c)) { i++; continue; }
        if ((c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z')) {
            size_t j = i;
            while (j < n && (rust_alnum((unsigned char)src[j]) || src[j] == '_')) j++;
            const std::string id = src.substr(i, j - i);
            i = j;
            Token t;
            if (id == "true" || id == "yes") { t.kind = Token::KBool; t.b = true; }
            else if (id == "false" || id == "no") { t.kind = Token::KBool; t.b = false; }
            else { t.kind = Token::KIdent; t.text = id; }
            out.push_back(t);
            continue;
        }
        if (c == '"') {   // tokenize.rs:331-351 (the escape flag survives an "\n", as there)
            i++;
            bool escape = false;
            std::string s;
            while (i < n) {
                const char ch = src[i++];
                if (ch == '"' && !escape) break;
                if (ch == '\\' && !escape) { escape = true; continue; }
                if (ch == 'n' && escape) { s.push_back('\n'); continue; }
                s.push_back(ch);
                escape = false;
            }
            Token t; t.kind = Token::KString; t.text = s;
            out.push_back(t);
            continue;
        }
        if (c >= '0' && c <= '9') { read_number(false); continue; }
        if (c == '#') { while (i < n && src[i] != '\n') i++; continue; }
        i++;   // every case below has consumed c
        auto peek = [&]() -> int { return i < n ? (unsigned char)src[i] : -1; };
        switch (c) {
            case '+': tok_op(Op::Add); break;
            case '-': {
                if (peek() < 0) fail("tokenizer error: unexpected end of input after -");
                if (peek() >= '0' && peek() <= '9') read_number(true); else tok_op(Op::Sub);
                break;
            }
            case '*': tok_op(Op::Mul); break;
            case '/': tok_op(Op::Div); break;
            case '%': tok_op(Op::Mod); break;
            case '<': if (peek() == '=') { i++; tok_op(Op::LtEq); } else if (peek() == '<') { i++; tok_op(Op::Push); } else tok_op(Op::Lt); break;
            case '>': if (peek() == '=') { i++; tok_op(Op::GtEq); } else tok_op(Op::Gt); break;
            case ',': tok_sep(Sep::Comma); break;
            case '{': tok_sep(Sep::BraceOpen); break;
            case '}': tok_sep(Sep::BraceClose); break;
            case '(': tok_sep(Sep::ParensOpen); break;
            case ')': tok_sep(Sep::ParensClose); break;
            case '[': tok_sep(Sep::BracketOpen); break;
            case ']': tok_sep(Sep::BracketClose); break;
            case ':': tok_sep(Sep::Colon); break;
            case '=': if (peek() == '=') { i++; tok_op(Op::Eq); } else tok_op(Op::Assign); break;
            case '!': if (peek() == '=') { i++; tok_op(Op::Neq); } else tok_op(Op::Not); break;
            case '&':
                if (peek() == '&') { i++; tok_op(Op::And); }
                else fail(std::string("tokenizer error: unexpected character ") + (peek() < 0 ? "<end>" : std::string(1, (char)peek())));
                break;
            case '|':   // tokenize.rs:297-305 pushes Op::And here too
                if (peek() == '|') { i++; tok_op(Op::And); }
                else fail(std::string("tokenizer error: unexpected character ") + (peek() < 0 ? "<end>" : std::string(1, (char)peek())));
                break;
            case '.':
                if (peek() == '.') { i++; tok_op(Op::Range); }
                else fail(std::string("tokenizer error: unexpected character ") + (peek() < 0 ? "<end>" : std::string(1, (char)peek())));
                break;
            default: fail(std::string("tokenizer error: unexpected character ") + (char)c);
        }
    }
    return out;
}

// ------------------------------------------------------------------ syntax tree (ast.rs:61-185)
struct Node;
using NodeP = std::shared_ptr<Node>;
using Props = std::vector<std::pair<std::string, NodeP>>;   // later duplicates win, as in a HashMap built from a list
struct Node {
    enum K { Object, Assign, For, If, Function, Return, Dict, Call, Ident, Str, Num, Vec, Bool, Array, ScopeEnd, Index, Bin } k;
    std::string name;                 // Object, Assign, For (variable), Function, Call, Ident, Str
    double num = 0.;
    bool flag = false;                // Assign: declare; Bool: value
    Op op = Op::Add;                  // Bin
    std::vector<NodeP> kids;          // Call args, Array items, Vec x y z, Bin a b, Index (array, index), Assign/Return value, For (from, to)
    Props props;                      // Object, Dict
    std::vector<std::string> params;  // Function
    std::vector<NodeP> body;          // For, Function
    std::vector<std::pair<NodeP, std::vector<NodeP>>> conds;   // If
    bool has_else = false;
    std::vector<NodeP> else_body;
    explicit Node(K kind) : k(kind) {}
};
NodeP mk(Node::K k) { return std::make_shared<Node>(k); }

struct ExcessClose { NodeP first; };   // ast.rs:57-58: how a call's last argument reports the closing parenthesis

int precedence(Op o) {   // ast.rs:17-31
    switch (o) {
        case Op::Mul: case Op::Div: case Op::Mod: return 3;
        case Op::Add: case Op::Sub: return 2;
        case Op::Eq: case Op::Neq: case Op::Gt: case Op::Lt: case Op::GtEq: case Op::LtEq: return 1;
        default: return 0;
    }
}
bool arithmetic(Op o) { return o == Op::Add || o == Op::Sub || o == Op::Mul || o == Op::Div || o == Op::Mod; }

class Parser {
   public:
    explicit Parser(std::vector<Token> t) : toks_(std::move(t)) {}

    std::vector<NodeP> parse_root() {   // ast.rs:221-231
        std::vector<NodeP> body = parse_scope();
        if (!body.empty() && body.back()->k == Node::ScopeEnd) fail("ast error: expected 'not a closing brace', got '}'");
        return body;
    }

   private:
    std::vector<Token> toks_;
    size_t pos_ = 0;

    const Token* peek() const { return pos_ < toks_.size() ? &toks_[pos_] : nullptr; }
    Token next() {
        if (pos_ >= toks_.size()) fail("ast error: expected more tokens, got end");
        return toks_[pos_++];
    }
    void expect_sep(Sep s) {
        Token t = next();
        if (!t.is(s)) { Token w; w.kind = Token::KSep; w.sep = s; fail("ast error: expected '" + show(w) + "', got '" + show(t) + "'"); }
    }
    void expect_op(Op o) {
        Token t = next();
        if (!t.is(o)) { Token w; w.kind = Token::KOp; w.op = o; fail("ast error: expected '" + show(w) + "', got '" + show(t) + "'"); }
    }
    void expect_ident(const char* s) {
        Token t = next();
        if (!t.is_ident(s)) fail(std::string("ast error: expected '") + s + "', got '" + show(t) + "'");
    }
    std::string read_ident() {
        Token t = next();
        if (t.kind != Token::KIdent) fail("ast error: expected 'an identifier', got '" + show(t) + "'");
        return t.text;
    }
    std::vector<NodeP> closed_scope() {
        std::vector<NodeP> body = parse_scope();
        if (body.empty() || body.back()->k != Node::ScopeEnd) fail("ast error: expected more tokens, got end");
        return body;
    }

    // a comma-separated list of values up to `close`; the last value may swallow a closing parenthesis itself
    std::vector<NodeP> value_list(Sep close) {   // ast.rs:424-457, 602-635, 670-703
        std::vector<NodeP> v;
        while (true) {
            if (peek() && peek()->is(close)) { next(); break; }
            NodeP item;
            bool more = true;
            try { item = parse_value(true); }
            catch (ExcessClose& e) {
                if (!e.first) fail("ast error: too many closing parenthesis");
                item = e.first; more = false;
            }
            v.push_back(item);
            if (!more) break;
            if (peek() && peek()->is(close)) { next(); break; }
            expect_sep(Sep::Comma);
        }
        return v;
    }

    std::vector<NodeP> parse_scope() {   // ast.rs:234-484
        std::vector<NodeP> nodes;
        while (pos_ < toks_.size()) {
            Token token = next();
            if (token.kind == Token::KIdent) {
                const std::string& id = token.text;
                if (id == "for") {
                    NodeP f = mk(Node::For);
                    f->name = read_ident();
                    expect_ident("in");
                    f->kids.push_back(parse_value(true));
                    expect_ident("to");
                    f->kids.push_back(parse_value(true));
                    expect_sep(Sep::BraceOpen);
                    f->body = closed_scope();
                    nodes.push_back(f);
                    continue;
                }
                if (id == "if") {
                    NodeP n = mk(Node::If);
                    NodeP cond = parse_value(true);
                    expect_sep(Sep::BraceOpen);
                    n->conds.emplace_back(cond, closed_scope());
                    while (peek() && peek()->is_ident("else")) {
                        next();
                        Token t = next();
                        if (t.is_ident("if")) {
                            NodeP c = parse_value(true);
                            expect_sep(Sep::BraceOpen);
                            n->conds.emplace_back(c, closed_scope());
                        } else if (t.is(Sep::BraceOpen)) {
                            n->else_body = closed_scope();
                            n->has_else = true;
                            break;
                        } else fail("ast error: expected '`if` or opening brace', got '" + show(t) + "'");
                    }
                    nodes.push_back(n);
                    continue;
                }
                if (id == "let") {
                    NodeP a = mk(Node::Assign);
                    a->name = read_ident();
                    a->flag = true;
                    expect_op(Op::Assign);
                    a->kids.push_back(parse_value(true));
                    nodes.push_back(a);
                    continue;
                }
                if (id == "fn") {
                    NodeP f = mk(Node::Function);
                    f->name = read_ident();
                    expect_sep(Sep::ParensOpen);
                    while (true) {   // ast.rs:941-978 with identifiers as items
                        if (peek() && peek()->is(Sep::ParensClose)) { next(); break; }
                        f->params.push_back(read_ident());
                        if (peek() && peek()->is(Sep::ParensClose)) { next(); break; }
                        expect_sep(Sep::Comma);
                    }
                    expect_sep(Sep::BraceOpen);
                    f->body = closed_scope();
                    nodes.push_back(f);
                    continue;
                }
                if (id == "return") {
                    NodeP r = mk(Node::Return);
                    r->kids.push_back(parse_value(true));
                    nodes.push_back(r);
                    continue;
                }
                const Token* p = peek();
                if (!p) continue;
                if (p->is(Op::Assign)) {
                    next();
                    NodeP a = mk(Node::Assign);
                    a->name = id;
                    a->kids.push_back(parse_value(true));
                    nodes.push_back(a);
                } else if (p->is(Sep::BraceOpen)) {
                    next();
                    NodeP o = mk(Node::Object);
                    o->name = id;
                    o->props = read_dict();
                    nodes.push_back(o);
                } else if (p->is(Sep::ParensOpen)) {
                    next();
                    NodeP c = mk(Node::Call);
                    c->name = id;
                    c->kids = value_list(Sep::ParensClose);
                    nodes.push_back(c);
                } else {
                    fail("ast error: expected 'something valid in a scope', got '" + show(next()) + "'");
                }
            } else if (token.is(Sep::BraceClose)) {
                nodes.push_back(mk(Node::ScopeEnd));
                return nodes;
            } else {
                fail("ast error: expected 'something usable in a scope, or a scope terminator', got '" + show(token) + "'");
            }
        }
        return nodes;
    }

    // The shunting-yard expression reader of ast.rs:487-830.  `logic` = comparison and logic operators belong to the
    // expression (false inside a <x, y, z> literal, whose `>` must end it).
    NodeP parse_value(bool logic) {
        struct StackItem { bool paren; Op op; };
        std::vector<StackItem> ops;
        std::vector<NodeP> out;
        auto apply = [&](Op o) {
            if (!logic && !arithmetic(o)) fail("ast error: error parsing arithmetic expression");
            if (out.size() < 2) fail("ast error: operator without two operands");
            NodeP b = out.back(); out.pop_back();
            NodeP a = out.back(); out.pop_back();
            NodeP n = mk(Node::Bin);
            n->op = o; n->kids = {a, b};
            out.push_back(n);
        };
        bool last_op = true;
        while (true) {
            const Token* p = peek();
            if (!p) fail("ast error: expected more tokens, got end");
            if (p->kind == Token::KNumber || p->kind == Token::KString || p->kind == Token::KBool) {
                if (!last_op) break;
                last_op = false;
                Token t = next();
                NodeP n = mk(t.kind == Token::KNumber ? Node::Num : t.kind == Token::KString ? Node::Str : Node::Bool);
                n->num = t.num; n->name = t.text; n->flag = t.b;
                out.push_back(n);
            } else if (p->is(Sep::BraceOpen)) {
                if (!last_op) break;
                last_op = false;
                next();
                NodeP d = mk(Node::Dict);
                d->props = read_dict();
                out.push_back(d);
            } else if (p->is(Sep::BracketOpen)) {
                if (out.empty()) {
                    if (!last_op) break;
                    last_op = false;
                    next();
                    NodeP a = mk(Node::Array);
                    a->kids = value_list(Sep::BracketClose);
                    out.push_back(a);
                } else {
                    next();
                    NodeP index = parse_value(true);
                    expect_sep(Sep::BracketClose);
                    NodeP n = mk(Node::Index);
                    n->kids = {out.back(), index};
                    out.pop_back();
                    out.push_back(n);
                }
            } else if (p->is(Op::Lt) && last_op) {
                last_op = false;
                next();
                NodeP v = mk(Node::Vec);   // ast.rs:910-919
                v->kids.push_back(parse_value(false));
                expect_sep(Sep::Comma);
                v->kids.push_back(parse_value(false));
                expect_sep(Sep::Comma);
                v->kids.push_back(parse_value(false));
                expect_op(Op::Gt);
                out.push_back(v);
            } else if (p->kind == Token::KIdent) {
                if (!last_op) break;
                last_op = false;
                Token t = next();
                if (peek() && peek()->is(Sep::ParensOpen)) {
                    next();
                    NodeP c = mk(Node::Call);
                    c->name = t.text;
                    c->kids = value_list(Sep::ParensClose);
                    out.push_back(c);
                } else {
                    NodeP n = mk(Node::Ident);
                    n->name = t.text;
                    out.push_back(n);
                }
            } else if (p->kind == Token::KOp) {
                const Op o = p->op;
                const bool logic_op = o == Op::Eq || o == Op::Neq || o == Op::Lt || o == Op::Gt || o == Op::LtEq || o == Op::GtEq || o == Op::And || o == Op::Or;
                if (!(arithmetic(o) || (logic && logic_op))) break;
                last_op = true;
                next();
                while (!ops.empty() && !ops.back().paren) {
                    const int top = precedence(ops.back().op), cur = precedence(o);
                    bool pop = top > cur;
                    if (!pop && top == cur) {
                        // ast.rs:748-749 looks the new operator up in a table that only holds the arithmetic ones
                        if (!arithmetic(o)) fail("ast error: comparison / logic operators of equal precedence cannot be chained (the reference panics here); use parentheses");
                        pop = true;   // all of them are left-associative
                    }
                    if (!pop) break;
                    apply(ops.back().op);
                    ops.pop_back();
                }
                ops.push_back({false, o});
            } else if (p->is(Sep::ParensOpen)) {
                last_op = true;
                next();
                ops.push_back({true, Op::Add});
            } else if (p->is(Sep::ParensClose)) {
                last_op = true;
                next();
                while (ops.empty() || !ops.back().paren) {
                    if (ops.empty()) throw ExcessClose{out.empty() ? nullptr : out.front()};
                    apply(ops.back().op);
                    ops.pop_back();
                }
                ops.pop_back();
            } else break;
        }
        while (!ops.empty()) {
            if (ops.back().paren) fail("ast error: error parsing arithmetic expression");
            apply(ops.back().op);
            ops.pop_back();
        }
        if (out.empty()) fail("ast error: expected a value");
        return out.front();
    }

    Props read_dict() {   // ast.rs:875-902 (the opening brace has been read)
        Props d;
        while (true) {
            if (peek() && peek()->is(Sep::BraceClose)) { next(); break; }
            Token t = next();
            if (t.kind != Token::KIdent) fail("ast error: expected 'a key-value or closing brace', got '" + show(t) + "'");
            if (peek() && peek()->is(Sep::Colon)) {
                next();
                d.emplace_back(t.text, parse_value(true));
            } else {
                NodeP n = mk(Node::Ident);   // `{ key }` is `{ key: key }`
                n->name = t.text;
                d.emplace_back(t.text, n);
            }
            if (peek() && peek()->is(Sep::BraceClose)) { next(); break; }
            expect_sep(Sep::Comma);
        }
        return d;
    }
};

// ------------------------------------------------------------------ values (interpret.rs:116-144)
struct Value;
using ArrayP = std::shared_ptr<std::vector<Value>>;
struct Value {
    enum K { Unit, String, Number, Vector, Color, Boolean, Dictionary, Array } k = Unit;
    std::string s;
    double n = 0.;
    Vector3 v;
    rth::Color c;
    bool b = false;
    std::shared_ptr<std::map<std::string, Value>> d;
    ArrayP a;   // arrays are reference objects: copies of the value share the storage (interpret.rs:209-216)
    static Value num(double x) { Value r; r.k = Number; r.n = x; return r; }
    static Value str(std::string x) { Value r; r.k = String; r.s = std::move(x); return r; }
    static Value vec(Vector3 x) { Value r; r.k = Vector; r.v = x; return r; }
    static Value col(rth::Color x) { Value r; r.k = Color; r.c = x; return r; }
    static Value boolean(bool x) { Value r; r.k = Boolean; r.b = x; return r; }
};
bool equal(const Value& a, const Value& b) {   // derived PartialEq
    if (a.k != b.k) return false;
    switch (a.k) {
        case Value::Unit: return true;
        case Value::String: return a.s == b.s;
        case Value::Number: return a.n == b.n;
        case Value::Vector: return a.v.x == b.v.x && a.v.y == b.v.y && a.v.z == b.v.z;
        case Value::Color: return a.c.r == b.c.r && a.c.g == b.c.g && a.c.b == b.c.b;
        case Value::Boolean: return a.b == b.b;
        case Value::Dictionary: {
            if (a.d->size() != b.d->size()) return false;
            for (auto& kv : *a.d) { auto it = b.d->find(kv.first); if (it == b.d->end() || !equal(kv.second, it->second)) return false; }
            return true;
        }
        default: return a.a == b.a;   // references: the same array object
    }
}
bool truthy(const Value& v) {   // interpret.rs:356-363
    if (v.k == Value::Boolean) return v.b;
    if (v.k == Value::Number) return v.n != 0. && !std::isnan(v.n);
    return v.k != Value::Unit;
}

uint8_t as_u8(double x) { return std::isnan(x) ? 0 : x <= 0. ? 0 : x >= 255. ? 255 : (uint8_t)x; }
int32_t as_i32(double x) { return std::isnan(x) ? 0 : x <= -2147483648. ? INT32_MIN : x >= 2147483647. ? INT32_MAX : (int32_t)x; }
uint32_t as_u32(double x) { return std::isnan(x) ? 0 : x <= 0. ? 0 : x >= 4294967295. ? UINT32_MAX : (uint32_t)x; }
size_t as_usize(double x) { return std::isnan(x) ? 0 : x <= 0. ? 0 : x >= 1.8446744073709552e19 ? SIZE_MAX : (size_t)x; }

std::string show_number(double x) {   // Rust's Display for f64: shortest digits that round-trip, never an exponent
    if (std::isnan(x)) return "NaN";
    if (std::isinf(x)) return x < 0 ? "-inf" : "inf";
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof(buf), x, std::chars_format::scientific);   // shortest round-trip digits
    std::string sci(buf, r.ptr), digits;
    const size_t e = sci.find('e');
    const int exp10 = std::atoi(sci.c_str() + e + 1);
    bool neg = false;
    for (size_t i = 0; i < e; i++) { if (sci[i] == '-') neg = true; else if (sci[i] != '.') digits.push_back(sci[i]); }
    while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
    std::string out;
    const int point = exp10 + 1;   // digits before the decimal point
    if (point <= 0) out = "0." + std::string((size_t)(-point), '0') + digits;
    else if ((size_t)point >= digits.size()) out = digits + std::string((size_t)point - digits.size(), '0');
    else out = digits.substr(0, (size_t)point) + "." + digits.substr((size_t)point);
    if (digits == "0") out = "0";
    return neg ? "-" + out : out;
}
std::string debug_value(const Value& v) {   // {:?} of the values print() can meet inside an array
    switch (v.k) {
        case Value::Unit: return "Unit";
        case Value::String: return "String(\"" + v.s + "\")";
        case Value::Number: { std::string s = show_number(v.n); if (s.find('.') == std::string::npos && std::isfinite(v.n)) s += ".0"; return "Number(" + s + ")"; }
        case Value::Vector: return "Vector(<" + show_number(v.v.x) + ", " + show_number(v.v.y) + ", " + show_number(v.v.z) + ">)";
        case Value::Color: return "Color(" + std::to_string(v.c.r) + ", " + std::to_string(v.c.g) + ", " + std::to_string(v.c.b) + ")";
        case Value::Boolean: return std::string("Boolean(") + (v.b ? "true" : "false") + ")";
        case Value::Dictionary: return "Dictionary(..)";
        default: { std::string s = "["; for (size_t i = 0; i < v.a->size(); i++) s += (i ? ", " : "") + debug_value((*v.a)[i]); return s + "]"; }
    }
}

// ------------------------------------------------------------------ gradient noise (stands in for the `noise` crate)
struct Noise {
    uint8_t p[512];
    Noise() {
        uint8_t base[256];
        for (int i = 0; i < 256; i++) base[i] = (uint8_t)i;
        uint64_t s = 0x9E3779B97F4A7C15ull;
        for (int i = 255; i > 0; i--) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            std::swap(base[i], base[s % (uint64_t)(i + 1)]);
        }
        for (int i = 0; i < 512; i++) p[i] = base[i & 255];
    }
    static double fade(double t) { return t * t * t * (t * (t * 6. - 15.) + 10.); }
    static double grad(int h, double x, double y, double z) {
        h &= 15;
        const double u = h < 8 ? x : y, v = h < 4 ? y : (h == 12 || h == 14) ? x : z;
        return ((h & 1) ? -u : u) + ((h & 2) ? -v : v);
    }
    double get(double x, double y, double z) const {
        const double fx = std::floor(x), fy = std::floor(y), fz = std::floor(z);
        const int X = (int)((long long)fx & 255), Y = (int)((long long)fy & 255), Z = (int)((long long)fz & 255);
        x -= fx; y -= fy; z -= fz;
        const double u = fade(x), v = fade(y), w = fade(z);
        const int A = p[X] + Y, AA = p[A] + Z, AB = p[A + 1] + Z, B = p[X + 1] + Y, BA = p[B] + Z, BB = p[B + 1] + Z;
        auto mix = [](double t, double a, double b) { return a + t * (b - a); };
        return mix(w, mix(v, mix(u, grad(p[AA], x, y, z), grad(p[BA], x - 1, y, z)),
                             mix(u, grad(p[AB], x, y - 1, z), grad(p[BB], x - 1, y - 1, z))),
                      mix(v, mix(u, grad(p[AA + 1], x, y, z - 1), grad(p[BA + 1], x - 1, y, z - 1)),
                             mix(u, grad(p[AB + 1], x, y - 1, z - 1), grad(p[BB + 1], x - 1, y - 1, z - 1))));
    }
};

// ------------------------------------------------------------------ interpreter (interpret.rs:403-1530)
struct UserFn { std::vector<std::string> params; std::vector<NodeP> body; };
struct Scope { std::map<std::string, Value> vars; std::map<std::string, UserFn> funcs; };
enum class Kind { Any, Dictionary, String, Number, Vector, Color, Boolean, Array };
bool matches(const Value& v, Kind k) {   // interpret.rs:366-380
    switch (k) {
        case Kind::Any: return true;
        case Kind::Dictionary: return v.k == Value::Dictionary;
        case Kind::String: return v.k == Value::String;
        case Kind::Number: return v.k == Value::Number;
        case Kind::Vector: return v.k == Value::Vector;
        case Kind::Color: return v.k == Value::Color;
        case Kind::Boolean: return v.k == Value::Boolean;
        default: return v.k == Value::Array;
    }
}

class Interp {
   public:
    Interp(const SdlOptions& o, Scene& s, std::string& printed) : opt_(o), scene_(s), printed_(printed), rng_(o.seed) {
        Scope g;   // interpret.rs:435-445
        g.vars["PI"] = Value::num(3.14159265358979323846);
        g.vars["TAU"] = Value::num(6.28318530717958647692);
        g.vars["E"] = Value::num(2.71828182845904523536);
        g.vars["t"] = Value::num(o.t);
        scopes_.push_back(std::move(g));
        build_functions();
    }
    void run(const std::vector<NodeP>& root) { run_scope(root); }

   private:
    using Args = std::vector<Value>;
    struct Builtin { std::vector<std::string> names; std::vector<Kind> kinds; std::function<Value(Interp&, Args&)> fn; };

    const SdlOptions& opt_;
    Scene& scene_;
    std::string& printed_;
    std::mt19937_64 rng_;
    Noise noise_;
    std::vector<Scope> scopes_;
    std::vector<std::string> object_names_;
    std::vector<Builtin> builtins_;
    std::map<std::string, std::shared_ptr<Texture>> images_;   // interpret.rs:397-398

    static const Value& arg(Args& v, size_t i) {
        if (i >= v.size()) fail("invalid args to function call (too few arguments: the reference panics here)");
        return v[i];
    }
    void emit(const std::string& line) {
        printed_ += line; printed_ += '\n';
        if (opt_.echo) std::printf("%s\n", line.c_str());
    }

    void build_functions() {   // interpret.rs:1156-1362, in that order (the first matching overload wins)
        using K = Kind;
        auto add = [&](std::vector<std::string> names, std::vector<Kind> kinds, std::function<Value(Interp&, Args&)> fn) {
            builtins_.push_back({std::move(names), std::move(kinds), std::move(fn)});
        };
        auto N = [](Args& v, size_t i) { return arg(v, i).n; };
        auto V = [](Args& v, size_t i) { return arg(v, i).v; };
        add({"add"}, {K::Number, K::Number}, [=](Interp&, Args& v) { return Value::num(N(v, 0) + N(v, 1)); });
        add({"sub"}, {K::Number, K::Number}, [=](Interp&, Args& v) { return Value::num(N(v, 0) - N(v, 1)); });
        add({"mul"}, {K::Number, K::Number}, [=](Interp&, Args& v) { return Value::num(N(v, 0) * N(v, 1)); });
        add({"div"}, {K::Number, K::Number}, [=](Interp&, Args& v) { return Value::num(N(v, 0) / N(v, 1)); });
        add({"mod"}, {K::Number, K::Number}, [=](Interp&, Args& v) { return Value::num(std::fmod(N(v, 0), N(v, 1))); });
        add({"pow"}, {K::Number, K::Number}, [=](Interp&, Args& v) { return Value::num(std::pow(N(v, 0), N(v, 1))); });
        add({"add"}, {K::Vector, K::Vector}, [=](Interp&, Args& v) { return Value::vec(V(v, 0) + V(v, 1)); });
        add({"sub"}, {K::Vector, K::Vector}, [=](Interp&, Args& v) { return Value::vec(V(v, 0) - V(v, 1)); });
        add({"mul"}, {K::Vector, K::Vector}, [=](Interp&, Args& v) { Vector3 a = V(v, 0), b = V(v, 1); return Value::vec({a.x * b.x, a.y * b.y, a.z * b.z}); });
        add({"div"}, {K::Vector, K::Vector}, [=](Interp&, Args& v) { Vector3 a = V(v, 0), b = V(v, 1); return Value::vec({a.x / b.x, a.y / b.y, a.z / b.z}); });
        add({"mul"}, {K::Vector, K::Number}, [=](Interp&, Args& v) { return Value::vec(V(v, 0) * N(v, 1)); });
        add({"div"}, {K::Vector, K::Number}, [=](Interp&, Args& v) { return Value::vec(V(v, 0) / N(v, 1)); });
        add({"add"}, {K::String, K::String}, [](Interp&, Args& v) { return Value::str(arg(v, 0).s + arg(v, 1).s); });
        add({"push"}, {K::Array, K::Any}, [](Interp&, Args& v) { arg(v, 0).a->push_back(arg(v, 1)); return Value(); });
        add({"set"}, {K::Array, K::Number, K::Any}, [](Interp&, Args& v) {
            std::vector<Value>& a = *arg(v, 0).a;
            const size_t i = as_usize(arg(v, 1).n);
            if (i == a.size()) a.push_back(arg(v, 2));
            else if (i < a.size()) a[i] = arg(v, 2);
            else fail("index out of bounds in set() (the reference panics here)");
            return Value();
        });
        add({"len"}, {K::Array}, [](Interp&, Args& v) { return Value::num((double)arg(v, 0).a->size()); });
        add({"print"}, {K::String}, [](Interp& s, Args& v) { s.emit(arg(v, 0).s); return Value(); });
        add({"print"}, {K::Number}, [](Interp& s, Args& v) { s.emit(show_number(arg(v, 0).n)); return Value(); });
        add({"print"}, {K::Boolean}, [](Interp& s, Args& v) { s.emit(arg(v, 0).b ? "true" : "false"); return Value(); });
        add({"print"}, {K::Array}, [](Interp& s, Args& v) { s.emit(debug_value(arg(v, 0))); return Value(); });
        add({"color", "rgb"}, {K::Number, K::Number, K::Number}, [=](Interp&, Args& v) { return Value::col(Color(as_u8(N(v, 0)), as_u8(N(v, 1)), as_u8(N(v, 2)))); });
        add({"gray"}, {K::Number}, [=](Interp&, Args& v) { const uint8_t c = as_u8(N(v, 0)); return Value::col(Color(c, c, c)); });
        add({"hsv"}, {K::Number, K::Number, K::Number}, [=](Interp&, Args& a) {   // interpret.rs:1258-1290
            const double h = std::fmod(N(a, 0), 360.), s = N(a, 1), v = N(a, 2);
            const double c = v * s, x = c * (1. - std::fabs(std::fmod(h / 60., 2.) - 1.)), m = v - c;
            double r, g, b;
            if (h >= 0. && h < 60.) { r = c; g = x; b = 0.; }
            else if (h >= 60. && h < 120.) { r = x; g = c; b = 0.; }
            else if (h >= 120. && h < 180.) { r = 0.; g = c; b = x; }
            else if (h >= 180. && h < 240.) { r = 0.; g = x; b = c; }
            else if (h >= 240. && h < 300.) { r = x; g = 0.; b = c; }
            else if (h >= 300. && h < 360.) { r = c; g = 0.; b = x; }
            else fail("hsv(): hue outside [0, 360) (the reference panics here)");
            return Value::col(Color(as_u8((r + m) * 255.), as_u8((g + m) * 255.), as_u8((b + m) * 255.)));
        });
        auto f1 = [&](const char* name, double (*fn)(double)) { add({name}, {K::Number}, [=](Interp&, Args& v) { return Value::num(fn(N(v, 0))); }); };
        f1("sin", std::sin); f1("cos", std::cos); f1("tan", std::tan); f1("asin", std::asin); f1("acos", std::acos); f1("atan", std::atan);
        f1("abs", std::fabs); f1("sqrt", std::sqrt); f1("floor", std::floor); f1("ceil", std::ceil);
        f1("rad", [](double x) { return x * (3.14159265358979323846 / 180.); });
        f1("deg", [](double x) { return x * (180. / 3.14159265358979323846); });
        add({"random"}, {K::Number, K::Number}, [=](Interp& s, Args& v) {
            const double lo = N(v, 0), hi = N(v, 1);
            if (!(lo <= hi)) fail("random(): empty range (the reference panics here)");
            const double u = (double)(s.rng_() >> 11) * (1. / 9007199254740992.);   // [0, 1)
            return Value::num(lo + (hi - lo) * u);
        });
        add({"lerp"}, {K::Number, K::Number, K::Number}, [=](Interp&, Args& v) { return Value::num(N(v, 0) + (N(v, 1) - N(v, 0)) * N(v, 2)); });
        add({"min"}, {K::Number, K::Number}, [=](Interp&, Args& v) { return Value::num(std::fmin(N(v, 0), N(v, 1))); });
        add({"max"}, {K::Number, K::Number}, [=](Interp&, Args& v) { return Value::num(std::fmax(N(v, 0), N(v, 1))); });
        add({"remap"}, {K::Number, K::Number, K::Number, K::Number, K::Number}, [=](Interp&, Args& v) {   // math/mod.rs:12-14
            return Value::num((N(v, 0) - N(v, 1)) * ((N(v, 4) - N(v, 3)) / (N(v, 2) - N(v, 1))) + N(v, 3));
        });
        add({"normalize"}, {K::Vector}, [=](Interp&, Args& v) { return Value::vec(normalize(V(v, 0))); });
        add({"magnitude"}, {K::Vector}, [=](Interp&, Args& v) { return Value::num(magnitude(V(v, 0))); });
        add({"angle"}, {K::Vector}, [=](Interp&, Args& v) {   // vector.rs:111-113
            const Vector3 a = V(v, 0), b = V(v, 1);
            if (arg(v, 1).k != Value::Vector) fail("angle(): the second argument must be a vector (the reference panics here)");
            return Value::num(std::acos((a.x * b.x + a.y * b.y + a.z * b.z) / (magnitude(a) * magnitude(b))));
        });
        add({"lerp"}, {K::Vector, K::Vector, K::Number}, [=](Interp&, Args& v) {
            const Vector3 a = V(v, 0), b = V(v, 1); const double t = N(v, 2);
            return Value::vec({a.x + (b.x - a.x) * t, a.y + (b.y - a.y) * t, a.z + (b.z - a.z) * t});
        });
        add({"perlin"}, {K::Number, K::Number}, [=](Interp& s, Args& v) { return Value::num(s.noise_.get(N(v, 0), N(v, 1), 0.)); });
        add({"perlin"}, {K::Number, K::Number, K::Number}, [=](Interp& s, Args& v) { return Value::num(s.noise_.get(N(v, 0), N(v, 1), N(v, 2))); });
        add({"simplex"}, {K::Number, K::Number}, [=](Interp& s, Args& v) { return Value::num(s.noise_.get(N(v, 0) + 31.7, N(v, 1) - 17.3, 5.1)); });
        add({"simplex"}, {K::Number, K::Number, K::Number}, [=](Interp& s, Args& v) { return Value::num(s.noise_.get(N(v, 0) + 31.7, N(v, 1) - 17.3, N(v, 2) + 5.1)); });
    }

    const Value* find_var(const std::string& name) const {   // interpret.rs:1496-1504
        for (size_t i = scopes_.size(); i-- > 0;) {
            auto it = scopes_[i].vars.find(name);
            if (it != scopes_[i].vars.end()) return &it->second;
        }
        return nullptr;
    }

    Value call(const std::string& name, const std::vector<NodeP>& arg_nodes) {   // interpret.rs:1129-1404
        Args values;
        for (auto& n : arg_nodes) values.push_back(eval(*n));
        for (auto& f : builtins_) {
            bool named = false;
            for (auto& nm : f.names) named = named || nm == name;
            if (!named) continue;
            bool ok = true;   // function.rs:36: only the common prefix of arguments and parameter kinds is compared
            for (size_t i = 0; i < values.size() && i < f.kinds.size(); i++) ok = ok && matches(values[i], f.kinds[i]);
            if (ok) return f.fn(*this, values);
        }
        for (size_t i = scopes_.size(); i-- > 0;) {
            auto it = scopes_[i].funcs.find(name);
            if (it == scopes_[i].funcs.end()) continue;
            const UserFn fn = it->second;
            Scope sc;
            for (size_t k = 0; k < fn.params.size() && k < values.size(); k++) sc.vars[fn.params[k]] = values[k];
            if (scopes_.size() > 2000) fail("function calls nested too deeply");
            scopes_.push_back(std::move(sc));
            Value ret = run_scope(fn.body);
            scopes_.pop_back();
            return ret;
        }
        fail("no function by the name " + name);
    }

    Value binary(const std::string& fn, const Node& n) { return call(fn, n.kids); }

    static int compare(const Value& a, const Value& b, bool& ordered) {   // cmp_inner!, interpret.rs:174-181
        ordered = true;
        if (a.k == Value::String && b.k == Value::String) return a.s < b.s ? -1 : a.s > b.s ? 1 : 0;
        if (a.k == Value::Number && b.k == Value::Number) {
            if (std::isnan(a.n) || std::isnan(b.n)) { ordered = false; return 0; }
            return a.n < b.n ? -1 : a.n > b.n ? 1 : 0;
        }
        if (a.k == Value::Boolean && b.k == Value::Boolean) return (int)a.b - (int)b.b;
        fail("bad comparison");
    }

    Value eval(const Node& n) {   // Value::from_node, interpret.rs:169-342
        switch (n.k) {
            case Node::Ident: {
                const Value* v = find_var(n.name);
                if (!v) fail("no variable defined by the name " + n.name);
                return *v;
            }
            case Node::Call: return call(n.name, n.kids);
            case Node::Str: return Value::str(n.name);
            case Node::Num: return Value::num(n.num);
            case Node::Bool: return Value::boolean(n.flag);
            case Node::Vec: {
                Value x = eval(*n.kids[0]), y = eval(*n.kids[1]), z = eval(*n.kids[2]);
                if (x.k != Value::Number || y.k != Value::Number || z.k != Value::Number) fail("vector components must be numbers (the reference panics here)");
                return Value::vec({x.n, y.n, z.n});
            }
            case Node::Dict: {
                Value d; d.k = Value::Dictionary; d.d = std::make_shared<std::map<std::string, Value>>();
                for (auto& kv : n.props) {   // entries that fail to evaluate are dropped (interpret.rs:203-207)
                    try { (*d.d)[kv.first] = eval(*kv.second); } catch (SdlError&) {}
                }
                return d;
            }
            case Node::Array: {
                Value a; a.k = Value::Array; a.a = std::make_shared<std::vector<Value>>();
                for (auto& item : n.kids) {
                    try { a.a->push_back(eval(*item)); } catch (SdlError&) {}
                }
                return a;
            }
            case Node::Index: {
                Value index = eval(*n.kids[1]);
                Value array = eval(*n.kids[0]);
                if (array.k != Value::Array) fail("attempt to index a variable that is not an array");
                if (index.k != Value::Number) fail("attempt to index an array with a non-number value");
                const size_t i = as_usize(index.n);
                if (i >= array.a->size()) fail("index out of bounds");
                return (*array.a)[i];
            }
            case Node::Bin: {
                switch (n.op) {
                    case Op::Add: return binary("add", n);
                    case Op::Sub: return binary("sub", n);
                    case Op::Mul: return binary("mul", n);
                    case Op::Div: return binary("div", n);
                    case Op::Mod: return binary("mod", n);
                    case Op::Eq: { Value a = eval(*n.kids[0]), b = eval(*n.kids[1]); return Value::boolean(equal(a, b)); }
                    case Op::Neq: { Value a = eval(*n.kids[0]), b = eval(*n.kids[1]); return Value::boolean(!equal(a, b)); }
                    case Op::Gt: case Op::Lt: case Op::GtEq: case Op::LtEq: {
                        Value a = eval(*n.kids[0]), b = eval(*n.kids[1]);
                        bool ordered;
                        const int c = compare(a, b, ordered);
                        if (!ordered) return Value::boolean(false);
                        return Value::boolean(n.op == Op::Gt ? c > 0 : n.op == Op::Lt ? c < 0 : n.op == Op::GtEq ? c >= 0 : c <= 0);
                    }
                    case Op::And: case Op::Or: {   // both sides are evaluated (interpret.rs:324-337)
                        Value a = eval(*n.kids[0]);
                        if (a.k != Value::Boolean) fail("attempt to perform logic on non-boolean types");
                        Value b = eval(*n.kids[1]);
                        if (b.k != Value::Boolean) fail("attempt to perform logic on non-boolean types");
                        return Value::boolean(n.op == Op::And ? (a.b && b.b) : (a.b || b.b));
                    }
                    default: break;
                }
                fail("cannot convert AST node to a definite type");
            }
            default: fail("cannot convert AST node to a definite type");
        }
    }

    // ---- object properties (interpret.rs:1447-1492): taken out of the map, evaluated, checked against a kind
    bool take(Props& props, const char* name, NodeP& out) {
        bool found = false;
        for (size_t i = 0; i < props.size();) {
            if (props[i].first == name) { out = props[i].second; found = true; props.erase(props.begin() + (long)i); }   // the last duplicate wins
            else i++;
        }
        return found;
    }
    bool has(const Props& props, const char* name) const {
        for (auto& kv : props) if (kv.first == name) return true;
        return false;
    }
    bool optional(Props& props, const char* name, Kind kind, Value& out) {
        NodeP node;
        if (!take(props, name, node)) return false;
        out = eval(*node);
        if (!matches(out, kind)) fail(std::string("invalid args to function call (property ") + name + ")");
        return true;
    }
    Value required(Props& props, const char* name, Kind kind) {
        Value v;
        if (!optional(props, name, kind, v)) fail(std::string("property ") + name + " must be specified (it is not optional)");
        return v;
    }
    bool already(const char* name) const {
        for (auto& n : object_names_) if (n == name) return true;
        return false;
    }

    std::string resolve(const std::string& path) const {
        if (opt_.base_dir.empty() || (!path.empty() && path[0] == '/')) return path;
        return opt_.base_dir + "/" + path;
    }
    std::shared_ptr<Texture> load_image(const std::string& file) {   // interpret.rs:687-694, 1109-1119
        auto it = images_.find(file);
        if (it != images_.end()) return it->second;
        uint32_t w = 0, h = 0;
        std::vector<uint8_t> rgb;
        std::string err;
        const std::string path = resolve(file);
        if (!read_image(path, w, h, rgb, err)) fail("generic image error: " + file + ": " + err);
        auto t = std::make_shared<Texture>(Texture::image(w, h, rgb.data()));
        images_[file] = t;
        return t;
    }

    Texture read_texture(const Node& node) {   // interpret.rs:1083-1125
        if (node.k != Node::Call) fail("invalid args to function call (texture)");
        Args v;
        for (auto& a : node.kids) v.push_back(eval(*a));
        auto want = [&](std::initializer_list<Kind> kinds) {
            if (v.size() != kinds.size()) fail("expected " + std::to_string(kinds.size()) + " args, got " + std::to_string(v.size()));
            size_t i = 0;
            for (Kind k : kinds) { if (!matches(v[i], k) || k == Kind::Array) fail("invalid args to function call (texture)"); i++; }
        };
        if (node.name == "solid") { want({Kind::Color}); return Texture::solid(v[0].c); }
        if (node.name == "checkerboard") { want({Kind::Color, Kind::Color}); return Texture::checkerboard(v[0].c, v[1].c); }
        if (node.name == "image") { want({Kind::String}); return *load_image(v[0].s); }
        fail("invalid args to function call (texture)");
    }

    Material read_material(Props& props) {   // interpret.rs:1047-1078
        NodeP node;
        if (!take(props, "material", node)) return Material();
        if (node->k != Node::Dict) fail("materials object must be a dictionary");
        Props map = node->props;
        Material m;
        Value v;
        m.reflectiveness = optional(map, "reflectiveness", Kind::Number, v) ? v.n : 0.;
        m.transparency = optional(map, "transparency", Kind::Number, v) ? v.n : 0.;
        m.ior = optional(map, "ior", Kind::Number, v) ? v.n : 1.5;
        m.emissivity = optional(map, "emissivity", Kind::Number, v) ? v.n : 0.;
        NodeP tex;
        m.texture = take(map, "texture", tex) ? read_texture(*tex) : Texture::solid(Color(255, 255, 255));
        return m;
    }

    void light_common(Props& p, Light& l) {
        Value v;
        if (optional(p, "color", Kind::Color, v)) l.color = v.c;
        if (optional(p, "intensity", Kind::Number, v)) l.intensity = v.n;
        if (optional(p, "specular_power", Kind::Number, v)) l.specular_power = as_i32(v.n);
        if (optional(p, "specular_strength", Kind::Number, v)) l.specular_strength = v.n;
    }

    // false: the object was skipped the way the reference's `continue` skips it (no entry in the object names)
    bool object(const std::string& name, Props props) {   // interpret.rs:604-1038
        Value v;
        if (name == "scene") {
            if (already("scene")) fail("there are too many definitions of the scene object, expecting zero or one");
            const bool have_depth = optional(props, "max_ray_depth", Kind::Number, v);
            const double depth = v.n;
            Value amb;
            const bool have_amb = optional(props, "ambient", Kind::Color, amb);
            if (have_depth) scene_.options.max_ray_depth = as_u32(depth);
            if (have_amb) scene_.options.ambient = amb.c;
        } else if (name == "camera") {
            if (already("camera")) fail("there are too many definitions of the camera object, expecting zero or one");
            Value vw, vh, origin, yaw, pitch, fov;
            const bool h_vw = optional(props, "vw", Kind::Number, vw), h_vh = optional(props, "vh", Kind::Number, vh);
            const bool h_o = optional(props, "origin", Kind::Vector, origin), h_y = optional(props, "yaw", Kind::Number, yaw);
            const bool h_p = optional(props, "pitch", Kind::Number, pitch), h_f = optional(props, "fov", Kind::Number, fov);
            if (h_vw) scene_.camera.vw = as_i32(vw.n);
            if (h_vh) scene_.camera.vh = as_i32(vh.n);
            if (h_o) scene_.camera.origin = origin.v;
            if (h_y) scene_.camera.yaw = yaw.n;
            if (h_p) scene_.camera.pitch = pitch.n;
            if (h_f) scene_.camera.set_fov(fov.n);
        } else if (name == "skybox") {
            if (already("skybox")) fail("there are too many definitions of the skybox object, expecting zero or one");
            const std::string t = required(props, "type", Kind::String).s;
            if (t == "normal") { scene_.skybox = Skybox(); scene_.skybox.kind = Skybox::Normal; }
            else if (t == "solid") { const Color c = required(props, "color", Kind::Color).c; scene_.skybox = Skybox(); scene_.skybox.kind = Skybox::Solid; scene_.skybox.color = c; }
            else if (t == "cubemap") {
                auto img = load_image(required(props, "image", Kind::String).s);
                if (img->w / 4 != img->h / 3 || img->w < 4) fail("cubemap: width/4 must equal height/3 (the reference asserts this, skybox.rs:49-51)");
                scene_.skybox = Skybox(); scene_.skybox.kind = Skybox::Cubemap; scene_.skybox.w = img->w; scene_.skybox.h = img->h; scene_.skybox.rgb = img->rgb;
            } else fail("materials object must be a dictionary");   // the reference reuses this error for an unknown skybox type
        } else if (name == "aabb" || name == "box") {
            SceneObject o; o.kind = SceneObject::KAabb;
            o.aabb.pos = required(props, "position", Kind::Vector).v;
            o.aabb.size = required(props, "size", Kind::Vector).v;
            o.aabb.material = read_material(props);
            scene_.objects.push_back(std::move(o));
        } else if (name == "mesh") {
            const Vector3 position = optional(props, "position", Kind::Vector, v) ? v.v : Vector3();
            const double scale = optional(props, "scale", Kind::Number, v) ? v.n : 1.;
            Value rxyz, rzyx;
            const bool h_xyz = optional(props, "rotate_xyz", Kind::Vector, rxyz), h_zyx = optional(props, "rotate_zyx", Kind::Vector, rzyx);
            Material material = read_material(props);
            auto mesh = std::make_shared<Mesh>();
            if (has(props, "obj")) {
                const std::string obj = required(props, "obj", Kind::String).s;
                std::string err;
                if (!Mesh::from_obj(resolve(obj), material, *mesh, err)) fail(err);
            } else {
                mesh->material = material;
                auto vectors = [](const Value& arr) { std::vector<Vector3> o; for (auto& e : *arr.a) if (e.k == Value::Vector) o.push_back(e.v); return o; };
                auto triples = [](const Value& arr) {
                    std::vector<uint32_t> flat;
                    for (auto& e : *arr.a) if (e.k == Value::Number) flat.push_back((uint32_t)std::min<size_t>(as_usize(e.n), UINT32_MAX));
                    std::vector<std::array<uint32_t, 3>> o;
                    for (size_t i = 0; i + 2 < flat.size(); i += 3) o.push_back({flat[i], flat[i + 1], flat[i + 2]});
                    return o;
                };
                mesh->verts = vectors(required(props, "verts", Kind::Array));
                const Value tris = required(props, "tris", Kind::Array);
                if (has(props, "normals")) {
                    mesh->normals = vectors(required(props, "normals", Kind::Array));
                    mesh->tri_normals = triples(required(props, "normal_indices", Kind::Array));
                }
                mesh->tris = triples(tris);
                if (mesh->tris.empty()) return false;
                for (auto& t : mesh->tris) for (uint32_t i : t) if (i >= mesh->verts.size()) fail("mesh: triangle index out of range (the reference panics here)");
            }
            if (scale != 1.) mesh->scale(scale);
            if (!optional(props, "recenter", Kind::Boolean, v) || v.b) mesh->center();
            if (h_xyz) {
                if (h_zyx) fail("property one of rotate_xyz, rotate_zyx, not duplicates must be specified (it is not optional)");
                mesh->rotate_xyz(rxyz.v);
            }
            if (h_zyx) mesh->rotate_zyx(rzyx.v);
            if (position != Vector3()) mesh->shift(position);
            if (mesh->normals.empty()) mesh->recalculate_normals();
            std::string err;
            if (!mesh->generate_sbvh(err)) fail(err);
            SceneObject o; o.kind = SceneObject::KMesh; o.mesh = mesh;
            scene_.objects.push_back(std::move(o));
        } else if (name == "plane") {
            SceneObject o; o.kind = SceneObject::KPlane;
            o.plane.origin = required(props, "origin", Kind::Vector).v;
            o.plane.normal = normalize(optional(props, "normal", Kind::Vector, v) ? v.v : Vector3(0., 1., 0.));
            o.plane.uv_wrap = optional(props, "uv_wrap", Kind::Number, v) ? (float)v.n : 1.f;
            o.plane.material = read_material(props);
            scene_.objects.push_back(std::move(o));
        } else if (name == "sphere") {
            SceneObject o; o.kind = SceneObject::KSphere;
            o.sphere.origin = required(props, "position", Kind::Vector).v;
            o.sphere.radius = required(props, "radius", Kind::Number).n;
            o.sphere.material = read_material(props);
            scene_.objects.push_back(std::move(o));
        } else if (name == "point_light" || name == "pointlight") {
            Light l = Light::point();
            light_common(props, l);
            l.vector = required(props, "position", Kind::Vector).v;
            if (optional(props, "max_distance", Kind::Number, v)) l.max_distance = v.n;
            scene_.lights.push_back(l);
        } else if (name == "sun" || name == "sun_light" || name == "sunlight") {
            Light l = Light::sun();
            light_common(props, l);
            l.vector = normalize(required(props, "vector", Kind::Vector).v);
            if (optional(props, "shadows", Kind::Boolean, v)) l.shadows = v.b;
            if (optional(props, "shadow_coefficient", Kind::Number, v)) l.shadow_coefficient = v.n;
            scene_.lights.push_back(l);
        } else if (name == "area_light" || name == "arealight") {
            Light l = Light::area();
            light_common(props, l);
            const std::string surface = required(props, "surface", Kind::String).s;
            if (surface == "sphere") {
                l.area_surface = RT_AREA_SPHERE;
                l.vector = required(props, "position", Kind::Vector).v;
                l.area_radius = required(props, "radius", Kind::Number).n;
            } else if (surface == "rectangle") {
                l.area_surface = RT_AREA_RECTANGLE;
                l.area_corners[0] = required(props, "c00", Kind::Vector).v;
                l.area_corners[1] = required(props, "c01", Kind::Vector).v;
                l.area_corners[2] = required(props, "c10", Kind::Vector).v;
                l.area_corners[3] = required(props, "c11", Kind::Vector).v;
            } else fail("materials object must be a dictionary");
            if (optional(props, "iterations", Kind::Number, v)) l.area_iterations = as_u32(v.n);
            if (optional(props, "max_distance", Kind::Number, v)) l.max_distance = v.n;
            scene_.lights.push_back(l);
        } else fail("unknown scene object/scene definition " + name);
        return true;
    }

    Value run_scope(const std::vector<NodeP>& nodes) {   // interpret.rs:509-1044
        for (auto& np : nodes) {
            const Node& n = *np;
            switch (n.k) {
                case Node::Assign: {
                    Value value = eval(*n.kids[0]);
                    if (n.flag) scopes_.back().vars[n.name] = value;
                    else {
                        bool done = false;   // the nearest scope that has the name, else the global scope
                        for (size_t i = scopes_.size(); i-- > 0 && !done;) {
                            auto it = scopes_[i].vars.find(n.name);
                            if (it != scopes_[i].vars.end()) { it->second = value; done = true; }
                            else if (i == 0) { scopes_[0].vars[n.name] = value; done = true; }
                        }
                    }
                    break;
                }
                case Node::For: {
                    Value from = eval(*n.kids[0]), to = eval(*n.kids[1]);
                    if (from.k != Value::Number || to.k != Value::Number) fail("for: the bounds must be numbers (the reference panics here)");
                    const int32_t a = as_i32(std::floor(from.n)), b = as_i32(std::floor(to.n));
                    for (int32_t i = a; i < b; i++) {
                        Scope sc;
                        sc.vars[n.name] = Value::num((double)i);
                        scopes_.push_back(std::move(sc));
                        run_scope(n.body);   // a `return` in here only ends this pass of the body
                        scopes_.pop_back();
                    }
                    break;
                }
                case Node::If: {
                    const std::vector<NodeP>* body = nullptr;
                    for (auto& cb : n.conds) if (truthy(eval(*cb.first))) { body = &cb.second; break; }
                    if (!body && n.has_else) body = &n.else_body;
                    if (body) {
                        scopes_.push_back(Scope());
                        run_scope(*body);
                        scopes_.pop_back();
                    }
                    break;
                }
                case Node::Function: scopes_.back().funcs[n.name] = UserFn{n.params, n.body}; break;
                case Node::Return: return eval(*n.kids[0]);
                case Node::Call: call(n.name, n.kids); break;
                case Node::Object: if (object(n.name, n.props)) object_names_.push_back(n.name); break;
                default: break;
            }
        }
        return Value();
    }
};

}  // namespace

bool load_sdl(const std::string& source, const SdlOptions& opt, Scene& scene, std::string& printed, std::string& err) {
    try {
        Parser parser(tokenize(source));
        std::vector<NodeP> root = parser.parse_root();
        Interp in(opt, scene, printed);
        in.run(root);
        return true;
    } catch (const SdlError& e) {
        err = e.msg;
    } catch (const ExcessClose&) {
        err = "ast error: too many closing parenthesis";
    } catch (const std::exception& e) {
        err = std::string("sdl: ") + e.what();
    }
    return false;
}

bool load_sdl_file(const std::string& path, const SdlOptions& opt, Scene& scene, std::string& printed, std::string& err) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { err = "cannot open " + path; return false; }
    std::stringstream ss;
    ss << f.rdbuf();
    return load_sdl(ss.str(), opt, scene, printed, err);
}

}  // namespace rth
