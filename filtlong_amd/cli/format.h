// format.h — numbers as the reference prints them (src/misc.cpp:24-49).  Included by main.cpp only.
#pragma once
#include <iomanip>
#include <locale>
#include <sstream>
#include <string>

static std::string double_to_string(double n) {  // src/misc.cpp:24-32
    std::stringstream ss;
    ss << std::fixed << std::setprecision(2) << n;
    std::string s = ss.str();
    if (s.size() < 5) return std::string(5 - s.size(), ' ') + s;
    return s;
}

static std::string int_to_string(long long n) {  // src/misc.cpp:35-40 (thousands grouping of the user's locale)
    std::stringstream ss;
    ss.imbue(std::locale(""));
    ss << std::fixed << n;
    return ss.str();
}

static std::string pad(const std::string &s, size_t width) { return width > s.size() ? s + std::string(width - s.size(), ' ') : s; }

