// args.h — the command line of the drop-in binary: flags, unit suffixes, validation order and messages of the reference
// (src/arguments.cpp:28-393, src/args.h for the generic readers).  Included by main.cpp only.
#pragma once
#include <sys/ioctl.h>
#include <climits>
#include <cmath>
#include <fstream>
#include <iostream>
#include <limits>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#define PROGRAM_VERSION "0.3.1"

// ------------------------------------------------------------------------------------------------ arguments
struct Args {
    std::string input_reads;
    bool target_bases_set = false; long long target_bases = 0;
    bool keep_percent_set = false; double keep_percent = 0;
    bool min_length_set = false; int min_length = 0;
    bool max_length_set = false; int max_length = 0;
    bool min_mean_q_set = false; double min_mean_q = 0;
    bool min_window_q_set = false; double min_window_q = 0;
    bool assembly_set = false; std::string assembly;
    std::vector<std::string> short_reads;
    double length_weight = 1.0, mean_q_weight = 1.0, window_q_weight = 1.0;
    bool trim = false;
    bool split_set = false; int split = 0;
    long long window_size = 250;
    bool verbose = false;
    int gpus = 1;  // not a reference flag: --gpus N scores on N GPUs of this node (one process per GPU)
};
enum ParsingResult { GOOD, BAD, HELP, VERSION };

struct ParseError : std::runtime_error { using std::runtime_error::runtime_error; };

static double read_double(const std::string &name, const std::string &value) {  // DoublesReader, arguments.cpp:28-39
    try {
        if (value.find_first_not_of("0123456789.") != std::string::npos) throw std::invalid_argument("");
        return std::stod(value);
    } catch (...) {
        throw ParseError("Error: argument '" + name + "' received invalid value type '" + value + "'");
    }
}

// "<number>[k|kb|m|mb|g|gb]", case-insensitive, fractional numbers allowed ("1.5k" = 1500), result truncated towards zero.
// Behaviour pinned by the reference's unit-suffix tests (test/test_unit_suffixes.py; arguments.cpp:53-93): anything but digits
// and dots after the optional sign starts the suffix, so "1e3" is an unknown suffix there and here.
static long long parse_int_with_suffix(const std::string &value) {
    const char *text = value.c_str();
    if (*text == '\0') throw std::invalid_argument("empty");
    // the numeric part: an optional sign, then digits and dots only (no exponent once a suffix follows)
    size_t i = (text[0] == '-') ? 1 : 0;
    const size_t digits_from = i;
    while (isdigit((unsigned char)text[i]) || text[i] == '.') ++i;
    if (text[i] == '\0') {  // no suffix: the whole token is the number (std::stod semantics, like the reference)
        size_t used = 0;
        const double v = std::stod(value, &used);
        return static_cast<long long>(v);
    }
    if (i == digits_from) throw std::invalid_argument("no number");
    double scale = 0.0;
    switch (tolower((unsigned char)text[i])) {
        case 'k': scale = 1e3; break;
        case 'm': scale = 1e6; break;
        case 'g': scale = 1e9; break;
        default: throw std::invalid_argument("suffix");
    }
    const char *rest = text + i + 1;
    if (!(rest[0] == '\0' || (tolower((unsigned char)rest[0]) == 'b' && rest[1] == '\0'))) throw std::invalid_argument("suffix");
    const double v = std::stod(value.substr(0, i));
    return static_cast<long long>(v * scale);
}

static long long read_ll_suffix(const std::string &name, const std::string &value) {  // arguments.cpp:42-51
    try { return parse_int_with_suffix(value); }
    catch (...) { throw ParseError("Error: argument '" + name + "' received invalid value '" + value + "'"); }
}

static int read_int_suffix(const std::string &name, const std::string &value) {  // arguments.cpp:96-113
    try {
        const long long r = parse_int_with_suffix(value);
        if (r > INT_MAX || r < INT_MIN) throw std::invalid_argument("Value out of range for int");
        return static_cast<int>(r);
    } catch (...) { throw ParseError("Error: argument '" + name + "' received invalid value '" + value + "'"); }
}

// The reference's default reader for --window_size (src/args.h:1609-1627): stream extraction into the flag's current value, and an
// error only if characters are left unread — so an empty value keeps the current one and a number beyond long long saturates.
static long long read_ll(const std::string &name, const std::string &value, long long current) {
    std::istringstream ss(value);
    long long v = current;
    ss >> v;
    if (ss.rdbuf()->in_avail() > 0) throw ParseError("Error: argument '" + name + "' received invalid value type '" + value + "'");
    return v;
}

// The help menu of the reference, byte for byte: src/arguments.cpp:126-221 declares the flags and groups and hands them to the
// formatter of the vendored src/args.h (Help(), 1065-1218, and Wrap(), 94-149), whose layout is a function of the width of the
// terminal on STDOUT (TIOCGWINSZ; indent 1 / 2 / 3 / 4 for widths up to 60 / 80 / 120 / beyond).  When stdout is no terminal the
// reference reads the width out of an untouched `struct winsize` — 0 in practice: the description comes out one word per line and,
// the other widths being unsigned differences that wrap around, nothing else is wrapped at all.  Both cases are restated here.
// (--gpus, this binary's one flag of its own, is documented in README.md: the menu is the reference's.)
static std::vector<std::string> help_wrap(const std::string &in, size_t width, size_t first = 0) {  // src/args.h:94-149
    std::vector<std::string> out;
    size_t cur = first ? first : width, linesize = 0;
    std::string line;
    std::istringstream ss(in);
    while (ss) {
        std::string item;
        ss >> item;
        if (linesize + 1 + item.size() > cur && linesize > 0) {
            out.push_back(line);
            line.clear();
            linesize = 0;
            cur = width;
        }
        if (!item.empty()) {
            if (linesize) { ++linesize; line += ' '; }
            line += item;
            linesize += item.size();
        }
    }
    if (linesize > 0) out.push_back(line);
    return out;
}

static void print_help(const char *prog) {
    struct winsize ws;
    memset(&ws, 0, sizeof ws);
    ioctl(STDOUT_FILENO, TIOCGWINSZ, &ws);
    const unsigned width = ws.ws_col;
    const unsigned indent = width > 120 ? 4 : width > 80 ? 3 : width > 60 ? 2 : 1;
    const unsigned flagindent = indent, eachgroup = indent, helpindent = 40, gutter = 1;  // progindent = descriptionindent = 0
    struct Entry { const char *names, *info; unsigned level; };
    static const Entry positional[] = {{"input_reads", "input long reads to be filtered", 0}};
    static const Entry optional[] = {
        {"output thresholds:", "", 0},
        {"-t[int], --target_bases [int]", "keep only the best reads up to this many total bases (unit suffixes: k, kb, m, mb, g, gb)", 1},
        {"-p[float], --keep_percent [float]", "keep only this percentage of the best reads (measured by bases)", 1},
        {"-l[int], --min_length [int]", "minimum length threshold (unit suffixes: k, kb, m, mb, g, gb)", 1},
        {"-L[int], --max_length [int]", "maximum length threshold (unit suffixes: k, kb, m, mb, g, gb)", 1},
        {"-q[float], --min_mean_q [float]", "minimum mean quality threshold", 1},
        {"--min_window_q [float]", "minimum window quality threshold", 1},
        {"NLexternal references (if provided, read quality will be determined using these instead of from the Phred scores):", "", 0},
        {"-a[file], --assembly [file]", "reference assembly in FASTA format", 1},
        {"-1[file], --short_1 [file]", "reference short reads in FASTQ format", 1},
        {"-2[file], --short_2 [file]", "reference short reads in FASTQ format", 1},
        {"NLscore weights (control the relative contribution of each score to the final read score):", "", 0},
        {"--length_weight [float]", "weight given to the length score (default: 1)", 1},
        {"--mean_q_weight [float]", "weight given to the mean quality score (default: 1)", 1},
        {"--window_q_weight [float]", "weight given to the window quality score (default: 1)", 1},
        {"NLread manipulation:", "", 0},
        {"--trim", "trim non-k-mer-matching bases from start/end of reads", 1},
        {"--split [split]", "split reads at this many (or more) consecutive non-k-mer-matching bases (unit suffixes: k, kb, m, mb, g, gb)", 1},
        {"NLother:", "", 0},
        {"--window_size [int]", "size of sliding window used when measuring window quality (default: 250)", 1},
        {"--verbose", "verbose output to stderr with info for each read", 1},
        {"--version", "display the program version and quit", 1},
        {"-h, --help", "display this help menu", 0},
    };
    std::ostream &os = std::cerr;
    const std::string progline = std::string("usage: ") + prog + " {OPTIONS} [input_reads]";
    const auto proglines = help_wrap(progline, (size_t)(unsigned)(width - 4u), (size_t)width);
    for (size_t i = 0; i < proglines.size(); ++i) os << (i ? "    " : "") << proglines[i] << '\n';  // (progtailindent 4)
    os << '\n';
    for (const auto &l : help_wrap("Filtlong: a quality filtering tool for Nanopore and PacBio reads", width)) os << l << "\n";
    os << "\n";
    auto block = [&](const Entry &e, bool optional_rules) {
        const unsigned groupindent = e.level * eachgroup;
        const std::string names = e.names;
        const bool nl = optional_rules && names.compare(0, 2, "NL") == 0;
        // (unsigned differences, as in the reference: a width below the indents wraps around to "never wrap")
        const unsigned flag_width = nl ? width - (flagindent + gutter) : width - (flagindent + helpindent + gutter);
        const auto flags = help_wrap(names, (size_t)flag_width);
        const auto info = help_wrap(e.info, (size_t)(unsigned)(width - (helpindent + groupindent)));
        size_t flagssize = 0;
        for (size_t i = 0; i < flags.size(); ++i) {
            if (i) os << '\n';
            const bool line_nl = optional_rules && flags[i].compare(0, 2, "NL") == 0;
            if (optional_rules && (line_nl || flags[i].compare(0, 2, "-h") == 0)) os << '\n';
            os << std::string(groupindent + flagindent, ' ') << (line_nl ? flags[i].substr(2) : flags[i]);
            flagssize = flags[i].size() - (line_nl ? 2 : 0);
        }
        size_t k = 0;
        if (flagindent + flagssize + gutter > helpindent || info.empty()) {
            os << '\n';
        } else {
            os << std::string(helpindent - (flagindent + flagssize), ' ') << info[0] << '\n';
            k = 1;
        }
        for (; k < info.size(); ++k) os << std::string(groupindent + helpindent, ' ') << info[k] << '\n';
    };
    os << "positional arguments:\n";
    for (const Entry &e : positional) block(e, false);
    os << "\n";
    os << "optional arguments:\n";
    for (const Entry &e : optional) block(e, true);
    os << "\n";
    for (const auto &l : help_wrap("For more information, go to: https://github.com/rrwick/Filtlong", width)) os << l << "\n";
}

static bool file_exists(const std::string &f) { std::ifstream in(f); return in.good(); }

static ParsingResult parse_args(int argc, char **argv, Args &a) {
    bool version = false;
    bool short1_set = false, short2_set = false;
    std::string short1, short2;
    std::vector<std::string> positional;
    bool options_ended = false;
    try {
        for (int i = 1; i < argc; ++i) {
            std::string tok = argv[i];
            std::string flag, value;
            bool have_value = false;
            // (the reference's parser reports a second positional where it meets it, before anything behind it: src/args.h:1437)
            auto take_positional = [&]() {
                if (!positional.empty())
                    throw ParseError("Error: passed in argument, but no positional arguments were ready to receive it: " + tok);
                positional.push_back(tok);
            };
            if (options_ended) {
                take_positional();
                continue;
            }
            if (tok == "--") {  // everything behind it is positional (src/args.h: the terminator)
                options_ended = true;
                continue;
            }
            if (tok.size() >= 2 && tok[0] == '-' && tok[1] == '-') {
                flag = tok.substr(2);  // long flags take their value from the next token (LongSeparator(" "), arguments.cpp:128)
            } else if (tok.size() >= 2 && tok[0] == '-' && !(isdigit((unsigned char)tok[1]) && false)) {
                flag = std::string(1, tok[1]);
                if (tok.size() > 2) { value = tok.substr(2); have_value = true; }  // -t100
                static const char *shorts = "tplLqa12h";
                if (!strchr(shorts, tok[1])) throw ParseError("Error: flag could not be matched: '" + std::string(1, tok[1]) + "'");
            } else {
                take_positional();
                continue;
            }
            auto need = [&](const char *) -> std::string {
                if (have_value) return value;
                if (i + 1 >= argc) throw ParseError("Error: flag '" + flag + "' requires an argument but received none");
                return argv[++i];
            };
            if (flag == "h" || flag == "help") { print_help(argv[0]); return HELP; }
            else if (flag == "version") version = true;
            else if (flag == "verbose") a.verbose = true;
            else if (flag == "trim") a.trim = true;
            else if (flag == "t" || flag == "target_bases") { a.target_bases = read_ll_suffix("int", need("target_bases")); a.target_bases_set = true; }
            else if (flag == "p" || flag == "keep_percent") { a.keep_percent = read_double("float", need("keep_percent")); a.keep_percent_set = true; }
            else if (flag == "l" || flag == "min_length") { a.min_length = read_int_suffix("int", need("min_length")); a.min_length_set = true; }
            else if (flag == "L" || flag == "max_length") { a.max_length = read_int_suffix("int", need("max_length")); a.max_length_set = true; }
            else if (flag == "q" || flag == "min_mean_q") { a.min_mean_q = read_double("float", need("min_mean_q")); a.min_mean_q_set = true; }
            else if (flag == "min_window_q") { a.min_window_q = read_double("float", need("min_window_q")); a.min_window_q_set = true; }
            else if (flag == "a" || flag == "assembly") { a.assembly = need("assembly"); a.assembly_set = true; }
            else if (flag == "1" || flag == "short_1") { short1 = need("short_1"); short1_set = true; }
            else if (flag == "2" || flag == "short_2") { short2 = need("short_2"); short2_set = true; }
            else if (flag == "length_weight") a.length_weight = read_double("float", need("length_weight"));
            else if (flag == "mean_q_weight") a.mean_q_weight = read_double("float", need("mean_q_weight"));
            else if (flag == "window_q_weight") a.window_q_weight = read_double("float", need("window_q_weight"));
            else if (flag == "split") { a.split = read_int_suffix("split", need("split")); a.split_set = true; }
            else if (flag == "window_size") a.window_size = read_ll("int", need("window_size"), a.window_size);
            else if (flag == "gpus") a.gpus = (int)read_ll("int", need("gpus"), a.gpus);
            else throw ParseError("Error: flag could not be matched: " + flag);
        }
    } catch (const ParseError &e) {
        std::cerr << e.what() << "\n";
        return BAD;
    }
    if (argc == 1) { print_help(argv[0]); return HELP; }
    if (version) return VERSION;
    if (!positional.empty()) a.input_reads = positional[0];
    if (a.input_reads.empty()) { std::cerr << "Error: input reads are required" << "\n"; return BAD; }
    if (short1_set) a.short_reads.push_back(short1);
    if (short2_set) a.short_reads.push_back(short2);

    // validation: same order and messages as arguments.cpp:298-393
    const bool some_reference = !a.short_reads.empty() || a.assembly_set;
    if (a.trim && !some_reference) { std::cerr << "Error: assembly or read reference is required to use --trim" << "\n"; return BAD; }
    if (a.split_set && !some_reference) { std::cerr << "Error: assembly or read reference is required to use --split" << "\n"; return BAD; }
    std::vector<std::string> files;
    files.push_back(a.input_reads);
    for (auto &f : a.short_reads) files.push_back(f);
    if (a.assembly_set) files.push_back(a.assembly);
    for (auto &f : files)
        if (!file_exists(f)) { std::cerr << "Error: cannot find file: " << f << "\n"; return BAD; }
    if (!a.trim && !a.split_set && !a.target_bases_set && !a.keep_percent_set && !a.min_length_set && !a.max_length_set &&
        !a.min_mean_q_set && !a.min_window_q_set) {
        std::cerr << "Error: no thresholds set, you must use one of the following options:\n";
        std::cerr << "target_bases, keep_percent, min_length, max_length, min_mean_q, min_window_q, trim, split\n";
        return BAD;
    }
    if (a.target_bases_set && a.target_bases <= 0) { std::cerr << "Error: the value for --target_bases must be a positive integer\n"; return BAD; }
    if (a.min_length_set && a.min_length <= 0) { std::cerr << "Error: the value for --min_length must be a positive integer\n"; return BAD; }
    if (a.max_length_set && a.max_length <= 0) { std::cerr << "Error: the value for --max_length must be a positive integer\n"; return BAD; }
    if (a.keep_percent_set && (a.keep_percent <= 0.0 || a.keep_percent >= 100.0)) {
        std::cerr << "Error: the value for --keep_percent must be greater than 0 and less than 100\n"; return BAD; }
    if (a.min_mean_q_set && a.min_mean_q <= 0.0) { std::cerr << "Error: the value for --min_mean_q must be greater than 0\n"; return BAD; }
    if (a.min_window_q_set && a.min_window_q <= 0.0) { std::cerr << "Error: the value for --min_window_q must be greater than 0\n"; return BAD; }
    if (a.length_weight < 0.0 || a.mean_q_weight < 0.0 || a.window_q_weight < 0.0) { std::cerr << "Error: weight values cannot be negative\n"; return BAD; }
    if (a.split_set && a.split <= 0) { std::cerr << "Error: the value for --split must be a positive integer\n"; return BAD; }
    // The reference reads the option as a long long and stores it in an `int` field before validating it
    // (src/arguments.cpp:295,389; src/arguments.h:90): 4294967546 scores with a window of 250, 2147483648 is "not positive".
    // The drop-in narrows the same way, then validates.
    a.window_size = (long long)(int32_t)(uint32_t)(unsigned long long)a.window_size;
    if (a.window_size <= 0) { std::cerr << "Error: the value for --window_size must be a positive integer\n"; return BAD; }
    if (a.gpus < 1 || a.gpus > 64) { std::cerr << "Error: the value for --gpus must be between 1 and 64\n"; return BAD; }
    return GOOD;
}
