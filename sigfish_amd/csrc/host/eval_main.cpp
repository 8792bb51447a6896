// eval_main.cpp -- `sigfish-amd eval truth.paf test.paf`: mapping accuracy of a test PAF against a truth PAF by
// read id, the reference's `sigfish eval` (src/eval.c:380-445) with the same report on stdout.
// A test mapping is correct when target and strand agree with one of the read's truth mappings and either the
// starts or the ends lie within 100 bases (is_correct_overlap, src/eval.c:218-242).
#include <getopt.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

struct Paf {
    std::string rid, tid;
    int qlen = 0, qs = 0, qe = 0, tlen = 0, ts = 0, te = 0, mapq = 0;
    int strand = 0;
    char tp = 'P';
};

bool parse_paf(const std::string &line, Paf *p) {  // src/eval.c:73-146: 12 mandatory columns, then tags
    std::vector<std::string> f;
    size_t i = 0;
    while (i <= line.size()) {
        size_t e = line.find_first_of("\t\r\n", i);
        if (e == std::string::npos) e = line.size();
        if (e > i) f.push_back(line.substr(i, e - i));
        i = e + 1;
    }
    if (f.size() < 12) return false;
    p->rid = f[0];
    p->qlen = atoi(f[1].c_str());
    p->qs = atoi(f[2].c_str());
    p->qe = atoi(f[3].c_str());
    if (f[4] == "+")
        p->strand = 0;
    else if (f[4] == "-")
        p->strand = 1;
    else
        return false;
    p->tid = f[5];
    p->tlen = atoi(f[6].c_str());
    p->ts = atoi(f[7].c_str());
    p->te = atoi(f[8].c_str());
    p->mapq = atoi(f[11].c_str());
    p->tp = 'P';
    for (size_t k = 12; k < f.size(); ++k) {
        if (f[k] == "tp:A:P") p->tp = 'P';
        if (f[k] == "tp:A:S") p->tp = 'S';
    }
    return true;
}

bool correct(const Paf &a, const Paf &b, bool tid_only) {
    if (a.tid != b.tid || a.strand != b.strand) return false;
    if (tid_only) return true;
    const int ds = abs(a.ts - b.ts), de = abs(a.te - b.te);
    return (de < ds ? de : ds) < 100;
}

}  // namespace

int eval_main(int argc, char **argv) {
    static option lo[] = {{"verbose", required_argument, 0, 'v'}, {"help", no_argument, 0, 'h'},        {"version", no_argument, 0, 'V'},
                          {"output", required_argument, 0, 'o'},  {"secondary", required_argument, 0, 1}, {"tid-only", no_argument, 0, 2},
                          {0, 0, 0, 0}};
    bool sec = true, tid_only = false, help = false;
    int c, li = 0;
    optind = 1;
    while ((c = getopt_long(argc, argv, "o:hV", lo, &li)) >= 0) {
        if (c == 'V') {
            fprintf(stdout, "sigfish-amd eval\n");
            return 0;
        } else if (c == 'h') {
            help = true;
        } else if (c == 1) {
            sec = (!strcmp(optarg, "yes") || !strcmp(optarg, "y"));
        } else if (c == 2) {
            tid_only = true;
        }
    }
    if (argc - optind < 2 || help) {
        FILE *fp = help ? stdout : stderr;
        fprintf(fp, "Usage: sigfish-amd eval truth.paf test.paf\n\nbasic options:\n   -h                         help\n"
                    "   --version                  print version\n   --secondary STR            consider secondary mappings. yes or no.\n"
                    "   --tid-only                 consider reference name and strand only\n");
        return help ? 0 : 1;
    }
    std::ifstream truth(argv[optind]), test(argv[optind + 1]);
    if (!truth) {
        fprintf(stderr, "[sigfish-amd] ERROR: cannot open %s\n", argv[optind]);
        return 1;
    }
    if (!test) {
        fprintf(stderr, "[sigfish-amd] ERROR: cannot open %s\n", argv[optind + 1]);
        return 1;
    }
    std::unordered_map<std::string, std::vector<Paf>> h;
    std::string line;
    long truth_rec = 0;
    while (std::getline(truth, line)) {
        Paf p;
        if (!parse_paf(line, &p)) {
            fprintf(stderr, "[sigfish-amd] ERROR: malformed PAF line in %s\n", argv[optind]);
            return 1;
        }
        h[p.rid].push_back(p);
        ++truth_rec;
    }
    long n_test = 0, n_correct = 0, n_incorrect = 0, only_b = 0;
    long mq_c[61] = {0}, mq_i[61] = {0};
    while (std::getline(test, line)) {
        Paf p;
        if (!parse_paf(line, &p)) {
            fprintf(stderr, "[sigfish-amd] ERROR: malformed PAF line in %s\n", argv[optind + 1]);
            return 1;
        }
        auto it = h.find(p.rid);
        if (it == h.end()) {
            ++only_b;
        } else {
            bool ok = false;
            for (const Paf &t : it->second)
                if ((sec || t.tp == p.tp) && correct(t, p, tid_only)) {
                    ok = true;
                    break;
                }
            if (p.mapq < 0 || p.mapq > 60) {
                fprintf(stderr, "[sigfish-amd] ERROR: mapq %d out of range\n", p.mapq);
                return 1;
            }
            if (ok) {
                ++n_correct;
                ++mq_c[p.mapq];
            } else {
                ++n_incorrect;
                ++mq_i[p.mapq];
            }
        }
        ++n_test;
    }
    fprintf(stderr, "Total mappings in testset: %ld\n", n_test);
    const long truth_mapped = static_cast<long>(h.size());
    printf("\nComparison between truthset and testset\nmapped_truthset\t%ld\nmapped_testset\t%ld (%.2f%%)\ncorrect\t%ld (%.2f%%)\n"
           "incorrect\t%ld (%.2f%%)\nonly_in_testset\t%ld\n",
           truth_mapped, n_test, n_test / static_cast<float>(truth_mapped) * 100, n_correct, n_correct / static_cast<float>(n_test) * 100,
           n_incorrect, n_incorrect / static_cast<float>(n_test) * 100, only_b);
    printf("\n#mapq\tcorrect\tincorrect\n");
    for (int i = 60; i >= 0; --i)
        if (mq_c[i] || mq_i[i]) printf("%d\t%d\t%d\n", i, static_cast<int>(mq_c[i]), static_cast<int>(mq_i[i]));
    (void)truth_rec;
    return 0;
}
