/*
 * msa_oracle.cpp — CPU restatement of the reference MSA -> EDS / l-EDS transform.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Parity: pinned by the reference's
 * tests/cpp/test_msa.cpp vectors + SURVEY.md §8 KATs (reference MSA source needs SDSL,
 * absent here => unbuildable in this image; no oracle/_ref for this path).
 *
 * Structure follows the reference's three passes over the file image
 * (src/cpp/lib/transforms/msa_transforms.cpp):
 *   pass 1  parse_msa_and_build_variant_bv   :36-90
 *   pass 2a build_eds_boundaries             :101-115
 *   pass 2b build_leds_boundaries            :133-190
 *   pass 3  generate_output                  :200-324
 * The std::istream is replaced by a byte image + explicit offsets; getline / tellg /
 * seekg / read are emulated with their observable semantics (short read at EOF, the
 * read buffer that is reused without clearing, scan stop at '\0').
 */
#include "oracle.h"

#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

namespace {

struct MsaMeta {                       // msa_transforms.cpp:18-24
    std::string ref_seq;
    std::vector<int64_t> start_positions;
    size_t n_sequences = 0;
    size_t seq_length = 0;
    long line_width = -1;
};

// pass 1 — msa_transforms.cpp:36-90.  B has ref.size()+1 entries (sentinel at the end).
void pass1(const uint8_t* f, size_t n, MsaMeta& meta, std::vector<uint8_t>& B)
{
    uint64_t counter = 0;
    size_t i = 0;
    size_t pos = 0;
    bool have_B = false;
    while (pos < n) {                                        // std::getline loop :46
        const uint8_t* nl = static_cast<const uint8_t*>(memchr(f + pos, '\n', n - pos));
        size_t end = nl ? static_cast<size_t>(nl - f) : n;
        const uint8_t* line = f + pos;
        size_t len = end - pos;
        bool hit_eof = (nl == nullptr);
        pos = nl ? end + 1 : n;
        if (len == 0) continue;                              // :47-49
        if (line[0] == '>') {                                // :51
            if (counter == 1) {                              // :53-57
                B.assign(meta.ref_seq.size() + 1, 1);
                have_B = true;
            }
            i = 0;
            counter++;
            // tellg() after a getline that hit EOF without a delimiter returns -1
            meta.start_positions.push_back(hit_eof ? -1 : static_cast<int64_t>(pos));
        } else if (counter == 1) {                           // :62-68
            meta.ref_seq.append(reinterpret_cast<const char*>(line), len);
            if (meta.line_width == -1) meta.line_width = static_cast<long>(len);
        } else {                                             // :69-80
            if (!have_B) throw std::runtime_error("oracle: sequence data before first header");
            for (size_t j = 0; j < len; j++) {
                if (i >= meta.ref_seq.size())
                    throw std::runtime_error("oracle: sequence longer than the first one "
                                             "(out-of-bounds in the reference)");
                if (line[j] != static_cast<uint8_t>(meta.ref_seq[i]) || line[j] == '-') B[i] = 0;
                i++;
            }
        }
    }
    if (!have_B || meta.ref_seq.empty())
        throw std::runtime_error("oracle: MSA needs >= 2 sequences (undefined in the reference)");
    B[meta.ref_seq.size()] = B[meta.ref_seq.size() - 1] ^ 1; // :84
    meta.n_sequences = counter;
    meta.seq_length = meta.ref_seq.size();
}

// pass 2a — msa_transforms.cpp:101-115
std::vector<uint8_t> eds_boundaries(const std::vector<uint8_t>& B)
{
    std::vector<uint8_t> H(B.size(), 0);
    H[0] = 1;
    for (size_t i = 1; i < B.size(); i++)
        if (B[i] != B[i - 1]) H[i] = 1;
    return H;
}

// pass 2b — msa_transforms.cpp:133-190.  select_0/select_1 are only used to find the end
// of the current run; the sentinel bit guarantees a run end <= L exists.
std::vector<uint8_t> leds_boundaries(const std::vector<uint8_t>& B, size_t l, size_t L)
{
    std::vector<uint8_t> H(B.size(), 0);
    size_t i = 0;
    bool prev_was_standalone = false;
    while (i < L) {
        size_t e = i;
        while (e < B.size() && B[e] == B[i]) e++;            // next_zero / next_one
        if (B[i]) {
            size_t run_length = e - i;
            bool standalone = (run_length >= l || i == 0 || e == L);      // :153
            if (standalone) {
                H[i] = 1;
                prev_was_standalone = true;
            } else {
                if (prev_was_standalone) H[i] = 1;           // :161-163 (unreachable)
                prev_was_standalone = false;
            }
        } else {
            if (prev_was_standalone) {                       // :176-179
                H[i] = 1;
                prev_was_standalone = false;
            }
        }
        i = e;
    }
    H[0] = 1;                                                // :187
    return H;
}

// pass 3 — msa_transforms.cpp:200-324
void generate_output(const uint8_t* f, size_t n, const MsaMeta& meta,
                     const std::vector<uint8_t>& B, const std::vector<uint8_t>& H,
                     std::string& eds_out, std::string& seds_out)
{
    const size_t L = meta.ref_seq.size();
    std::vector<size_t> starts;                              // select_h
    for (size_t i = 0; i < L; i++) if (H[i]) starts.push_back(i);
    const size_t n_symbols = starts.size();                  // :213-216
    const size_t lw = static_cast<size_t>(meta.line_width);

    std::vector<char> buffer(meta.seq_length + (meta.seq_length / lw) + 10, 0);   // :219

    for (size_t sym = 0; sym < n_symbols; sym++) {
        size_t start_pos = starts[sym];
        size_t end_pos = (sym + 1 < n_symbols) ? starts[sym + 1] : L;
        size_t region_length = end_pos - start_pos;

        bool is_common = true;                               // :235-241
        for (size_t i = start_pos; i < end_pos; i++) if (B[i] == 0) { is_common = false; break; }

        eds_out.push_back('{');
        if (is_common) {                                     // :245-258
            for (size_t i = start_pos; i < end_pos; i++)
                if (meta.ref_seq[i] != '-') eds_out.push_back(meta.ref_seq[i]);
            seds_out += "{0}";
        } else {
            std::map<std::string, std::set<int>> variant_to_paths;
            std::vector<std::string> insertion_order;
            for (size_t s = 0; s < meta.n_sequences; s++) {  // :266-294
                int64_t file_pos = meta.start_positions[s] +
                                   static_cast<int64_t>(start_pos + (start_pos / lw));
                size_t tmp = ((start_pos % lw) + region_length) / lw;
                size_t bytes_to_read = region_length + tmp;
                // in.clear(); in.seekg(file_pos); in.read(buffer, bytes_to_read)
                if (file_pos >= 0 && static_cast<size_t>(file_pos) <= n) {
                    size_t avail = n - static_cast<size_t>(file_pos);
                    size_t got = bytes_to_read < avail ? bytes_to_read : avail;
                    memcpy(buffer.data(), f + file_pos, got);   // rest of buffer keeps stale bytes
                }
                std::string variant;
                for (size_t k = 0; k < bytes_to_read && buffer[k] != '\0'; k++)
                    if (buffer[k] != '\n' && buffer[k] != '-') variant.push_back(buffer[k]);
                int path_id = static_cast<int>(s) + 1;
                if (variant_to_paths.find(variant) == variant_to_paths.end())
                    insertion_order.push_back(variant);
                variant_to_paths[variant].insert(path_id);
            }
            for (size_t v = 0; v < insertion_order.size(); v++) {   // :297-317
                const std::string& variant = insertion_order[v];
                const std::set<int>& paths = variant_to_paths[variant];
                eds_out += variant;
                if (v + 1 < insertion_order.size()) eds_out.push_back(',');
                seds_out.push_back('{');
                size_t p = 0;
                for (int id : paths) {
                    seds_out += std::to_string(id);
                    if (++p < paths.size()) seds_out.push_back(',');
                }
                seds_out.push_back('}');
            }
        }
        eds_out.push_back('}');
    }
}

char* dup_out(const std::string& s, size_t* n)
{
    char* p = static_cast<char*>(malloc(s.size() + 1));
    memcpy(p, s.data(), s.size());
    p[s.size()] = 0;
    *n = s.size();
    return p;
}

void set_err(char* err, size_t cap, const char* what)
{
    if (err && cap) { strncpy(err, what, cap - 1); err[cap - 1] = 0; }
}

} // namespace

extern "C" int oracle_msa(const uint8_t* file, size_t n, uint32_t l,
                          char** eds, size_t* eds_n, char** seds, size_t* seds_n,
                          char* err, size_t errcap)
{
    try {
        MsaMeta meta;
        std::vector<uint8_t> B;
        pass1(file, n, meta, B);
        std::vector<uint8_t> H = (l == 0) ? eds_boundaries(B)               // :334-345
                                          : leds_boundaries(B, l, meta.ref_seq.size()); // :351-365
        std::string e, s;
        generate_output(file, n, meta, B, H, e, s);
        *eds = dup_out(e, eds_n);
        *seds = dup_out(s, seds_n);
        return 0;
    } catch (const std::exception& ex) {
        set_err(err, errcap, ex.what());
        return 2;
    }
}

/* Column slab [c0,c1) of the alignment transformed as a whole alignment (EDS mode).
 * Built by materialising the slab as its own single-line MSA image and running the
 * same three passes; used only to test the multi-GPU boundary stitch on CPU. */
extern "C" int oracle_msa_slab(const uint8_t* file, size_t n, uint64_t c0, uint64_t c1,
                               char** eds, size_t* eds_n, char** seds, size_t* seds_n,
                               char* err, size_t errcap)
{
    try {
        // gather rows (headers dropped, newlines inside rows dropped)
        std::vector<std::string> rows;
        size_t pos = 0;
        while (pos < n) {
            const uint8_t* nl = static_cast<const uint8_t*>(memchr(file + pos, '\n', n - pos));
            size_t end = nl ? static_cast<size_t>(nl - file) : n;
            if (end > pos) {
                if (file[pos] == '>') rows.emplace_back();
                else if (!rows.empty())
                    rows.back().append(reinterpret_cast<const char*>(file + pos), end - pos);
            }
            pos = nl ? end + 1 : n;
        }
        std::string img;
        for (size_t r = 0; r < rows.size(); r++) {
            if (c1 > rows[r].size() || c0 >= c1) throw std::runtime_error("oracle: bad slab range");
            img += ">r\n";
            img.append(rows[r], c0, c1 - c0);
            img.push_back('\n');
        }
        return oracle_msa(reinterpret_cast<const uint8_t*>(img.data()), img.size(), 0,
                          eds, eds_n, seds, seds_n, err, errcap);
    } catch (const std::exception& ex) {
        set_err(err, errcap, ex.what());
        return 2;
    }
}

extern "C" void oracle_free(void* p) { free(p); }
