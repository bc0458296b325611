/*
 * TEST INFRASTRUCTURE — never shipped, never built on the GPU box.
 *
 * The reference's own pipeline (VeyFastTreeImpl<float, SSE128Operations>::run, the instantiation its dispatcher picks for
 * nucleotides in single precision, VeryFastTree.cpp:59-66) behind the reference's own command line (main.cpp's cli(),
 * compiled where it lies), with ONE difference from the reference binary: `options.verbose` is set again after
 * VeryFastTree::settings() has run.  settings() forces verbose = 1 whenever threads > 1 (VeryFastTree.cpp:68-70), which
 * silences the `Join` lines of the NJ phase (NJ.tcc:2993-3001) — the only way to see the join order.  The NJ phase's
 * OpenMP regions (top-hit refreshes, out-distance passes, the sort) do not change its result (the NJ-only
 * tree is identical for 1/2/4/8 threads in normal mode), so this program prints the one-thread join order in a
 * fraction of the one-thread time.  oracle/gen_fixtures.py `c4t` checks that claim before using the trace: the first
 * lines must equal the `-threads 1` trace of the reference binary itself as far as that one exists.
 *
 * No reference source is copied: main.cpp and the headers are #included from $(REF), the instantiation is linked from
 * oracle/_build/ref_VeryFastTreeFloatSSE128.o (oracle/Makefile).
 *
 * The log sink: the reference writes its log through std::cerr, which flushes after every `<<` (one write() per fragment;
 * at -verbose 3 that is most of the run time of the reference binary).  run() takes any std::ostream, so this program hands
 * it a stream whose buffer keeps the lines that start with "Join" and drops the rest - same lines, same text, no system
 * call per fragment.
 *
 *   njtrace <verbose> <reference command line ...>  2> joins.txt  > tree.nwk
 *   e.g.  njtrace 3 -nt -noml -nome -nosupport -threads 1 in.fa
 */
#define main vft_reference_main
#include "main.cpp"
#undef main

#include "VeryFastTree.h"
#include "operations/SSE128Operations.h"

extern template class veryfasttree::VeyFastTreeImpl<float, veryfasttree::SSE128Operations>;

namespace {
    /* keeps complete lines that begin with "Join", written to stderr through stdio's buffer */
    class JoinLines : public std::streambuf {
    public:
        ~JoinLines() override { fflush(stderr); }

    protected:
        int_type overflow(int_type ch) override {
            if (ch == traits_type::eof()) return traits_type::not_eof(ch);
            put((char) ch);
            return ch;
        }

        std::streamsize xsputn(const char *s, std::streamsize n) override {
            for (std::streamsize i = 0; i < n; i++) put(s[i]);
            return n;
        }

    private:
        void put(char c) {
            if (c == '\n') {
                if (line.size() >= 4 && line.compare(0, 4, "Join") == 0) {
                    line.push_back('\n');
                    fwrite(line.data(), 1, line.size(), stderr);
                }
                line.clear();
            } else if (line.size() < 4 || line.compare(0, 4, "Join") == 0) {
                line.push_back(c);
            } else if (line.size() == 4) {
                line.push_back(c);   /* a fifth character marks "not a Join line": the rest is dropped */
            }
        }

        std::string line;
    };
}

int main(int argc, char **argv) {
    if (argc < 3) {
        std::cerr << "usage: njtrace <verbose> <reference arguments>" << std::endl;
        return 2;
    }
    const int verbose = atoi(argv[1]);
    argc--;
    argv++;
    veryfasttree::Options options;
    std::string name = veryfasttree::Constants::name, version = veryfasttree::Constants::version,
            flags = veryfasttree::Constants::compileFlags;
    std::vector<std::string> args(argv + 1, argv + argc);
    CLI::App app;
    cli(app, name, version, flags, options, args);
    basicCli(app, name, version, flags);
    CLI11_PARSE(app, argc, argv);
    if (options.nCodes != 4 || options.doublePrecision) {
        std::cerr << "njtrace: nucleotides in single precision only" << std::endl;
        return 2;
    }
    std::ifstream finput(options.inFileName);
    if (finput.fail()) {
        std::cerr << "njtrace: cannot read " << options.inFileName << std::endl;
        return 2;
    }
    veryfasttree::VeryFastTree ft(options);
    {
        std::ostringstream quiet;
        ft.settings(quiet);      /* private: compiled with -fno-access-control */
    }
    ft.configOpenMP();
    if (ft.options.extension != "SSE3") {
        std::cerr << "njtrace: the dispatcher chose " << ft.options.extension << ", expected SSE3" << std::endl;
        return 2;
    }
    ft.options.verbose = verbose;
    bxz::istream input(finput);
    static char errbuf[1 << 20];
    setvbuf(stderr, errbuf, _IOFBF, sizeof(errbuf));
    JoinLines sink;
    std::ostream joinlog(&sink);
    veryfasttree::VeyFastTreeImpl<float, veryfasttree::SSE128Operations>(ft.options, input, std::cout, joinlog).run();
    std::cout.flush();
    fflush(stderr);
    return 0;
}
