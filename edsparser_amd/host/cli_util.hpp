// cli_util.hpp — minimal command-line option parser for the three tools.  The reference uses
// Boost.program_options (absent in this image); this accepts the same spellings:
//   --name value | --name=value | -n value | -nvalue | boolean switches, and reports errors with
//   Boost's wording so scripts that grep the messages keep working.
#pragma once
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace edsparser::cli {

struct Option {
    std::string long_name; char short_name; bool takes_value; bool required; std::string help;
};

class Parser {
public:
    explicit Parser(std::string caption) : caption_(std::move(caption)) {}
    void add(const std::string& long_name, char short_name, bool takes_value, bool required, const std::string& help)
    {
        opts_.push_back({long_name, short_name, takes_value, required, help});
    }
    void parse(int argc, char** argv)
    {
        for (int i = 1; i < argc; ++i) {
            std::string a = argv[i];
            const Option* o = nullptr;
            std::string value;
            bool have_value = false;
            if (a.rfind("--", 0) == 0) {
                std::string name = a.substr(2);
                size_t eq = name.find('=');
                if (eq != std::string::npos) { value = name.substr(eq + 1); name = name.substr(0, eq); have_value = true; }
                o = find_long(name);
                if (!o) throw std::runtime_error("unrecognised option '--" + name + "'");
            } else if (a.size() >= 2 && a[0] == '-') {
                o = find_short(a[1]);
                if (!o) throw std::runtime_error("unrecognised option '" + a + "'");
                if (a.size() > 2) { value = a.substr(2); have_value = true; }
            } else {
                throw std::runtime_error("too many positional options have been specified on the command line");
            }
            if (o->takes_value) {
                if (!have_value) {
                    if (i + 1 >= argc)
                        throw std::runtime_error("the required argument for option '--" + o->long_name + "' is missing");
                    value = argv[++i];
                }
                values_[o->long_name] = value;
            } else {
                values_[o->long_name] = "1";
            }
        }
    }
    // Boost's po::notify: required options must be present
    void notify() const
    {
        for (const auto& o : opts_)
            if (o.required && !values_.count(o.long_name))
                throw std::runtime_error("the option '--" + o.long_name + "' is required but missing");
    }
    bool has(const std::string& name) const { return values_.count(name) > 0; }
    std::string get(const std::string& name, const std::string& def = "") const
    {
        auto it = values_.find(name);
        return it == values_.end() ? def : it->second;
    }
    unsigned long get_unsigned(const std::string& name, unsigned long def) const
    {
        auto it = values_.find(name);
        if (it == values_.end()) return def;
        size_t used = 0;
        long long v = 0;
        try { v = std::stoll(it->second, &used); } catch (...) { used = 0; }
        if (used != it->second.size() || v < 0 || v > 0xffffffffll)
            throw std::runtime_error("the argument ('" + it->second + "') for option '--" + name + "' is invalid");
        return static_cast<unsigned long>(v);
    }
    long get_int(const std::string& name, long def) const
    {
        auto it = values_.find(name);
        if (it == values_.end()) return def;
        size_t used = 0;
        long v = 0;
        try { v = std::stol(it->second, &used); } catch (...) { used = 0; }
        if (used != it->second.size())
            throw std::runtime_error("the argument ('" + it->second + "') for option '--" + name + "' is invalid");
        return v;
    }
    std::string usage() const
    {
        std::string s = caption_ + ":\n";
        for (const auto& o : opts_) {
            std::string left = "  -" + std::string(1, o.short_name) + " [ --" + o.long_name + " ]";
            if (!o.short_name) left = "  --" + o.long_name;
            if (o.takes_value) left += " arg";
            if (left.size() < 34) left.append(34 - left.size(), ' '); else left += " ";
            s += left + o.help + "\n";
        }
        return s;
    }

private:
    const Option* find_long(const std::string& n) const { for (auto& o : opts_) if (o.long_name == n) return &o; return nullptr; }
    const Option* find_short(char c) const { for (auto& o : opts_) if (o.short_name && o.short_name == c) return &o; return nullptr; }
    std::string caption_;
    std::vector<Option> opts_;
    std::map<std::string, std::string> values_;
};

} // namespace edsparser::cli
