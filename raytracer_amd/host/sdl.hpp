// The scene description language of the reference's `sdl` crate (sdl/src/tokenize.rs, ast.rs, interpret.rs),
// rebuilt on top of the host Scene mirror so that scenes/*.sdl run unmodified.
//
// The language is defined by its implementation, so the behaviours a script can observe are kept, including the odd
// ones: `-` directly before a digit always starts a negative literal; `||` tokenizes to the same operator as `&&`
// (tokenize.rs:297-305); two values in a row end an expression, which is how statements are separated; `return`
// only leaves the innermost braces; both sides of `&&` are always evaluated; overloads match on a prefix of the
// argument list (function.rs:31-41).
//
// Deliberate differences, all about inputs the reference draws from outside the script:
//   * random(a, b) comes from a seeded generator (the reference uses an unseeded thread_rng, interpret.rs:1305);
//   * perlin()/simplex() are this library's own gradient noise (the reference calls the un-vendored `noise` 0.7
//     crate, interpret.rs:1341-1361): same range and character, different values;
//   * images are PNG or JPEG, decoded by this library's own readers (the reference uses the `image` crate).
#pragma once
#include <string>

#include "scene.hpp"

namespace rth {

struct SdlOptions {
    double t = 0.;            // global `t`: the frame number of --sequence (sdl/src/main.rs:84-92)
    uint64_t seed = 1;        // random()
    std::string base_dir;     // relative asset paths are resolved against it ("" = the working directory)
    bool echo = false;        // print() also writes to stdout
};

// Runs the script and builds the scene.  printed receives everything print() wrote, one line per call.
bool load_sdl(const std::string& source, const SdlOptions& opt, Scene& scene, std::string& printed, std::string& err);
bool load_sdl_file(const std::string& path, const SdlOptions& opt, Scene& scene, std::string& printed, std::string& err);

}  // namespace rth
