// utils.hpp -- string / number helpers of the netlist front-end.
//
// Same public names and semantics as the reference's include/utils.hpp
// (ltrim :4, toLower :10, parseSpiceNumber :20-74, isGroundName :76-79,
// clamp01 :80-84); the number parser decides the bits of every device
// parameter, so it keeps the reference's exact evaluation: std::stod on the
// lower-cased token, then ONE multiplication by the suffix factor.
#pragma once

#include <cctype>
#include <cstddef>
#include <string>

inline std::string ltrim(const std::string& s)
{
    const std::size_t first = s.find_first_not_of(" \t\r\n");
    return first == std::string::npos ? std::string() : s.substr(first);
}

inline std::string rtrim(const std::string& s)
{
    const std::size_t last = s.find_last_not_of(" \t\r\n");
    return last == std::string::npos ? std::string() : s.substr(0, last + 1);
}

inline std::string toLower(const std::string& s)
{
    std::string out(s);
    for (char& ch : out)
        ch = static_cast<char>(std::tolower(static_cast<unsigned char>(ch)));
    return out;
}

namespace csim_detail {
// SPICE magnitude suffixes understood by the reference (utils.hpp:34-44);
// anything else scales by 1.
inline double suffixFactor(const std::string& suf)
{
    struct Entry { const char* s; double f; };
    static const Entry table[] = {
        {"f", 1e-15}, {"p", 1e-12}, {"n", 1e-9}, {"u", 1e-6}, {"m", 1e-3},
        {"k", 1e3},   {"meg", 1e6}, {"g", 1e9},  {"t", 1e12},
    };
    for (const Entry& e : table)
        if (suf == e.s) return e.f;
    return 1.0;
}
} // namespace csim_detail

// "10k", "1u", "3e12", "3.3meg", ".25e-6" ...  Throws what std::stod throws
// when no numeric prefix exists (callers report and skip the statement).
inline double parseSpiceNumber(const std::string& token)
{
    const std::string s = toLower(token);
    std::size_t used = 0;
    double base = 0.0;
    bool parsed = true;
    try {
        base = std::stod(s, &used);
    } catch (...) {
        parsed = false;
    }
    if (parsed) {
        if (used == s.size()) return base;
        return base * csim_detail::suffixFactor(s.substr(used));
    }
    // std::stod refused the token: split at the first letter and retry.
    std::size_t cut = std::string::npos;
    for (std::size_t i = 0; i < s.size(); ++i) {
        if (std::isalpha(static_cast<unsigned char>(s[i]))) { cut = i; break; }
    }
    if (cut == std::string::npos) return 0.0;
    base = std::stod(s.substr(0, cut));
    return base * csim_detail::suffixFactor(s.substr(cut));
}

// node "0" / "gnd" (any case) is the reference node
inline bool isGroundName(const std::string& n)
{
    const std::string low = toLower(n);
    return low == "0" || low == "gnd";
}

inline double clamp01(double v)
{
    return v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
}
