// main.cpp — the `query` CLI: same flag surface and stdout contract as the reference's binary
// (query/src/main.rs:191-319), with the per-file scans running on the GPU(s).
#include <cstdio>

#include "pcq_host.hpp"

int main(int argc, char **argv) {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    return pcq::query_main(
        argc, argv,
        [](const std::string &s) {
            fputs(s.c_str(), stdout);
            fputc('\n', stdout);
        },
        [](const std::string &s) {
            fputs(s.c_str(), stderr);
            fputc('\n', stderr);
        });
}
