// gs_report.cpp -- MatchingResult.completeResults + ResultReporter.printMatchResult (C/match/MatchingResult.java:84-118,
// C/match/ResultReporter.java:190-279): the CSV of the match goal, with Java's Double.toString formatting.
#include "gs_ingest.h"

using namespace gs_host;

extern "C" int gs_host_java_double(double v, char *buf, int cap) {
    const std::string s = java_double(v);
    if (!buf || cap <= (int)s.size()) return GS_E_INVALID;
    memcpy(buf, s.c_str(), s.size() + 1);
    return GS_OK;
}

// ---------------------------------------------------------------------------------------------------
// completeResults + CSV
// ---------------------------------------------------------------------------------------------------
extern "C" int gs_host_write_csv(const char *path, const gs_host_tax_info *tax, const int64_t *table, const double *dtable,
                                 const gs_host_totals *totals) try {
    if (!path || !tax || !table || !totals || !tax->parent_vi || !tax->taxids || !tax->db_kmers)
        return hfail(GS_E_INVALID, "NULL argument");
    const int nv = tax->n_values;
    // rows: every value with a CountsPerTaxid (>= 1 hit k-mer or >= 1 classified read) plus all their ancestors
    // (MatchingResult.java:88-98)
    std::vector<char> present((size_t)nv, 0);
    for (int v = 0; v < nv; v++) {
        const int64_t *row = table + (size_t)v * GS_N_COLS;
        if (tax->parent_vi[v] != -2 && (row[GS_C_READS] > 0 || row[GS_C_READS_1KMER] > 0)) present[(size_t)v] = 1;
    }
    for (int v = 0; v < nv; v++)
        if (present[(size_t)v] == 1)
            for (int a = tax->parent_vi[v]; a >= 0; a = tax->parent_vi[a])
                if (!present[(size_t)a]) present[(size_t)a] = 2;
    // tree order (sortTaxidsViaTree): by position; default pre-order over children in value-index order
    std::vector<int> pos((size_t)nv, 0);
    if (tax->position) {
        for (int v = 0; v < nv; v++) pos[(size_t)v] = tax->position[v];
    } else {
        std::vector<std::vector<int>> kids((size_t)nv);
        std::vector<int> roots, stack;
        for (int v = 0; v < nv; v++) {
            if (tax->parent_vi[v] >= 0)
                kids[(size_t)tax->parent_vi[v]].push_back(v);
            else if (tax->parent_vi[v] == -1)
                roots.push_back(v);
        }
        int counter = 0;
        for (auto it = roots.rbegin(); it != roots.rend(); ++it) stack.push_back(*it);
        while (!stack.empty()) {
            const int v = stack.back();
            stack.pop_back();
            pos[(size_t)v] = counter++;
            for (auto it = kids[(size_t)v].rbegin(); it != kids[(size_t)v].rend(); ++it) stack.push_back(*it);
        }
    }
    std::vector<int> rows;
    for (int v = 0; v < nv; v++)
        if (present[(size_t)v]) rows.push_back(v);
    std::sort(rows.begin(), rows.end(), [&](int a, int b) { return pos[(size_t)a] < pos[(size_t)b]; });
    // accumulate into ancestors in tree order (:104-117); value types READS, KMERS, READS_BPS, READS_1KMER, READS_KMERS
    static const int vcol[5] = {GS_C_READS, GS_C_KMERS, GS_C_READS_BPS, GS_C_READS_1KMER, GS_C_READS_KMERS};
    static const char *vname[5] = {"reads", "kmers", "reads bps", "read >=1 kmer", "reads kmers"};
    std::vector<int64_t> acc((size_t)nv * 5, 0);
    std::vector<double> accn((size_t)nv * 5, 0.0), accd((size_t)nv * 4, 0.0);
    auto val = [&](int v, int t) { return present[(size_t)v] == 1 ? table[(size_t)v * GS_N_COLS + vcol[t]] : (int64_t)0; };
    auto dval = [&](int v, int j) { return (present[(size_t)v] == 1 && dtable) ? dtable[(size_t)v * GS_N_DCOLS + j] : 0.0; };
    for (int v : rows) {
        const int64_t dbk = tax->db_kmers[v];
        for (int t = 0; t < 5; t++) {
            acc[(size_t)v * 5 + t] += val(v, t);
            accn[(size_t)v * 5 + t] += dbk > 0 ? (double)val(v, t) / (double)dbk : 0.0;
        }
        for (int j = 0; j < 4; j++) accd[(size_t)v * 4 + j] += dval(v, j);
    }
    // a descendant adds its OWN values to every ancestor, in tree order
    for (int v : rows) {
        const int64_t dbk = tax->db_kmers[v];
        for (int a = tax->parent_vi[v]; a >= 0; a = tax->parent_vi[a]) {
            for (int t = 0; t < 5; t++) {
                acc[(size_t)a * 5 + t] += val(v, t);
                accn[(size_t)a * 5 + t] += dbk > 0 ? (double)val(v, t) / (double)dbk : 0.0;
            }
            for (int j = 0; j < 4; j++) accd[(size_t)a * 4 + j] += dval(v, j);
        }
    }
    FILE *f = fopen(path, "wb");
    if (!f) return hfail(GS_E_INVALID, std::string("cannot open ") + path);
    std::string o;
    o = "pos;level;name;rank;taxid;reads;kmers from reads;kmers;unique kmers;contigs;average contig length;max contig length;"
        "reads >=1 kmer;reads bps;avg. read length;db coverage;exp. unique kmers;unique kmers / exp.;db kmers;parent taxid;"
        "mean error;kmer error std. dev.;mean class error;class error std. dev.;contig len std. dev.;";
    for (int t = 0; t < 5; t++) o += std::string("norm. ") + vname[t] + ";";
    for (int t = 0; t < 5; t++) o += std::string("acc. ") + vname[t] + ";acc. norm. " + vname[t] + ";";
    o += "max contig desc.;acc. mean error;acc. error std. dev.;acc. mean class error;acc. class error std. dev.;";
    const int nmc = (tax->max_kmer_counts && tax->max_kmer_res_counts > 0) ? tax->max_kmer_res_counts : 0;
    if (nmc) o += "max kmer counts;";  // ResultReporter.java:213, :262-271
    o.push_back('\n');
    auto max_counts = [&](int row) {
        for (int i = 0; i < nmc; i++) {
            if (i > 0) o.push_back(';');
            append_int(o, tax->max_kmer_counts[(size_t)row * (size_t)nmc + (size_t)i]);
        }
        if (nmc) o.push_back(';');
    };
    auto dbl = [&](double v, bool total_row, bool always = false) {  // ResultReporter.java:249-253
        if (!std::isnan(v) && !std::isinf(v) && (!total_row || always)) o += java_double(v);
        o.push_back(';');
    };
    // TOTAL row (pos 0): reads, kmers, reads bps, db kmers; everything else 0 / blank
    o += "0;0;TOTAL;;;";
    append_int(o, totals->reads);
    o += ";0;";
    append_int(o, totals->kmers);
    o += ";0;0;";
    dbl(0.0 / 0.0, true);  // average contig length: NaN -> blank
    o += "0;0;";
    append_int(o, totals->bps);
    o.push_back(';');
    dbl(totals->reads ? (double)totals->bps / (double)totals->reads : 0.0 / 0.0, true, true);
    dbl(0, true);
    dbl(0, true);
    dbl(0, true);
    append_int(o, tax->db_kmers_total);
    o += ";;";
    for (int j = 0; j < 5; j++) dbl(0, true);
    for (int t = 0; t < 5; t++) dbl(0, true);
    for (int t = 0; t < 10; t++) o.push_back(';');
    o.push_back(';');
    for (int j = 0; j < 4; j++) dbl(0, true);
    max_counts(nv);
    o.push_back('\n');
    int p = 1;
    for (int v : rows) {
        const int64_t *row = table + (size_t)v * GS_N_COLS;
        const bool own = present[(size_t)v] == 1;
        auto col = [&](int c) { return own ? row[c] : (int64_t)0; };
        int level = 0;
        for (int a = tax->parent_vi[v]; a >= 0; a = tax->parent_vi[a]) level++;
        const int64_t reads = col(GS_C_READS), kmers = col(GS_C_KMERS), contigs = col(GS_C_CONTIGS);
        const int64_t uniq = own ? row[GS_C_UNIQUE_KMERS] : 0, dbk = tax->db_kmers[v];
        append_int(o, p++);
        o.push_back(';');
        append_int(o, level);
        o.push_back(';');
        if (tax->names && tax->names[v]) o += tax->names[v];
        o.push_back(';');
        if (tax->ranks && tax->ranks[v]) o += tax->ranks[v];
        o.push_back(';');
        o += tax->taxids[v];
        o.push_back(';');
        append_int(o, reads);
        o.push_back(';');
        append_int(o, col(GS_C_READS_KMERS));
        o.push_back(';');
        append_int(o, kmers);
        o.push_back(';');
        append_int(o, uniq);
        o.push_back(';');
        append_int(o, (int32_t)contigs);  // Java field is int
        o.push_back(';');
        dbl((double)kmers / (double)contigs, false);
        append_int(o, col(GS_C_MAX_CONTIG_LEN));
        o.push_back(';');
        append_int(o, col(GS_C_READS_1KMER));
        o.push_back(';');
        append_int(o, col(GS_C_READS_BPS));
        o.push_back(';');
        dbl((double)col(GS_C_READS_BPS) / (double)reads, false);
        dbl((double)uniq / (double)dbk, false);
        const double expu = (1 - std::pow(1 - 1.0 / (double)dbk, (double)kmers)) * (double)dbk;
        dbl(expu, false);
        dbl((double)uniq / expu, false);
        append_int(o, dbk);
        o.push_back(';');
        if (tax->parent_vi[v] >= 0) o += tax->taxids[tax->parent_vi[v]];
        o.push_back(';');
        const double es = dval(v, GS_D_ERR_SUM), es2 = dval(v, GS_D_ERR_SQ_SUM), cs = dval(v, GS_D_CLASS_ERR_SUM),
                     cs2 = dval(v, GS_D_CLASS_ERR_SQ_SUM);
        dbl(es / (double)reads, false);
        dbl(std::sqrt((es2 - es * es / (double)reads) / (double)(reads - 1)), false);
        dbl(cs / (double)reads, false);
        dbl(std::sqrt((cs2 - cs * cs / (double)reads) / (double)(reads - 1)), false);
        dbl(std::sqrt(((double)col(GS_C_CONTIG_LEN_SQ_SUM) - ((double)kmers * (double)kmers) / (double)contigs) / (double)(contigs - 1)), false);
        for (int t = 0; t < 5; t++) dbl((double)val(v, t) / (double)dbk, false);
        for (int t = 0; t < 5; t++) {
            append_int(o, acc[(size_t)v * 5 + t]);
            o.push_back(';');
            o += java_double(accn[(size_t)v * 5 + t]);
            o.push_back(';');
        }
        if (tax->max_contig_desc && tax->max_contig_desc[v]) o += tax->max_contig_desc[v];
        o.push_back(';');
        const double areads = (double)acc[(size_t)v * 5 + 0];
        const double aes = accd[(size_t)v * 4 + 0], aes2 = accd[(size_t)v * 4 + 1], acs = accd[(size_t)v * 4 + 2],
                     acs2 = accd[(size_t)v * 4 + 3];
        dbl(aes / areads, false);
        dbl(std::sqrt((aes2 - aes * aes / areads) / (areads - 1)), false);
        dbl(acs / areads, false);
        dbl(std::sqrt((acs2 - acs * acs / areads) / (areads - 1)), false);
        if (own) max_counts(v);
        else if (nmc) o.push_back(';');  // maxKMerCounts == null for rows added as missing ancestors
        o.push_back('\n');
    }
    // the CSV is the bit-exact deliverable: a short write (full disk, I/O error) must not pass for a result
    const bool written = fwrite(o.data(), 1, o.size(), f) == o.size();
    const bool closed = fclose(f) == 0;
    if (!written || !closed) return hfail(GS_E_IO, std::string("short write to ") + path);
    return GS_OK;
} catch (const std::bad_alloc &) {
    return hfail(GS_E_NOMEM, "out of host memory");
} catch (const std::exception &e) {  // (nothing may leave through the C ABI)
    return hfail(GS_E_INVALID, std::string("unexpected exception: ") + e.what());
}
