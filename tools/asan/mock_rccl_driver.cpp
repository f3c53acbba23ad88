// TEST INFRASTRUCTURE ONLY.  tests/mock_rccl/mock_rccl.cpp built with -DMOCK_RCCL_HOST_ONLY under AddressSanitizer / UBSan and
// driven as R forked processes on a CPU box: ring send / recv groups (messages below and above the 1 MiB ring, several per
// peer, crossing pairs), the all-to-neighbours pattern of a halo exchange, all-reduces, all-gathers and the mixed group the CG
// loop posts -- every value checked.  Exit code 0 = all ranks passed and no sanitizer report.
//     ./mock_rccl_asan [ranks = 8] [rounds = 6]
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <sys/wait.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(call) do { ncclResult_t r__ = (call); if (r__ != ncclSuccess) { fprintf(stderr, "rank %d: %s -> %s\n", rank, #call, ncclGetErrorString(r__)); return 1; } } while (0)

static double value(int src, int dst, int round, size_t i) { return src * 1000.0 + dst * 10.0 + round + 1e-6 * (double)(i % 9973); }

static int run_rank(int rank, int R, int rounds, ncclUniqueId id)
{
    ncclComm_t comm;
    CK(ncclCommInitRank(&comm, R, id, rank));
    hipStream_t st = nullptr;
    for (int round = 0; round < rounds; ++round) {
        // ---- every rank to every other rank: message sizes from 1 double to 1.5 ring lengths, two messages to the right neighbour
        const size_t big = round == 0 ? (size_t)(1.5 * (1u << 20) / 8) : 1000 + 377 * (size_t)round;
        std::vector<std::vector<double>> out(R), in(R);
        std::vector<double> out2(64), in2(64);
        CK(ncclGroupStart());
        for (int q = 0; q < R; ++q) {
            if (q == rank) continue;
            const size_t cnt = (q == (rank + 1) % R) ? big : 1 + (size_t)((rank * 7 + q * 3 + round) % 50);
            const size_t cin = (rank == (q + 1) % R) ? big : 1 + (size_t)((q * 7 + rank * 3 + round) % 50);
            out[q].resize(cnt); in[q].assign(cin, -1.0);
            for (size_t i = 0; i < cnt; ++i) out[q][i] = value(rank, q, round, i);
            CK(ncclSend(out[q].data(), cnt, ncclFloat64, q, comm, st));
            CK(ncclRecv(in[q].data(), cin, ncclFloat64, q, comm, st));
        }
        if (R > 1) {            // a second message on the same (src, dst) ring: issue order must be kept
            const int right = (rank + 1) % R, left = (rank + R - 1) % R;
            for (size_t i = 0; i < out2.size(); ++i) out2[i] = -value(rank, right, round, i);
            CK(ncclSend(out2.data(), out2.size(), ncclFloat64, right, comm, st));
            CK(ncclRecv(in2.data(), in2.size(), ncclFloat64, left, comm, st));
        }
        CK(ncclGroupEnd());
        for (int q = 0; q < R; ++q)
            for (size_t i = 0; i < in[q].size(); ++i)
                if (in[q][i] != value(q, rank, round, i)) { fprintf(stderr, "rank %d round %d: entry %zu from %d is %.17g\n", rank, round, i, q, in[q][i]); return 1; }
        if (R > 1)
            for (size_t i = 0; i < in2.size(); ++i)
                if (in2[i] != -value((rank + R - 1) % R, rank, round, i)) { fprintf(stderr, "rank %d round %d: second message out of order\n", rank, round); return 1; }
        // ---- the CG group: neighbour pairs + an all-reduce of two slots in ONE group
        double slots[2] = {1.0 + rank, 0.5}, halo_out = 7.0 + rank, halo_in = -1.0;
        CK(ncclGroupStart());
        if (R > 1) {
            CK(ncclSend(&halo_out, 1, ncclFloat64, (rank + 1) % R, comm, st));
            CK(ncclRecv(&halo_in, 1, ncclFloat64, (rank + R - 1) % R, comm, st));
        }
        CK(ncclAllReduce(slots, slots, 2, ncclFloat64, ncclSum, comm, st));
        CK(ncclGroupEnd());
        if (slots[0] != R * (R + 1) / 2.0 || slots[1] != 0.5 * R || (R > 1 && halo_in != 7.0 + (rank + R - 1) % R)) {
            fprintf(stderr, "rank %d round %d: mixed group gave %g %g %g\n", rank, round, slots[0], slots[1], halo_in);
            return 1;
        }
        // ---- all-gather of int32 tables (the `want` matrix of sgm_csr_create_dist) and of one wide row
        std::vector<int32_t> mine((size_t)R), all((size_t)R * R, -1);
        for (int q = 0; q < R; ++q) mine[q] = rank * 100 + q + round;
        CK(ncclAllGather(mine.data(), all.data(), (size_t)R, ncclInt32, comm, st));
        for (int p = 0; p < R; ++p)
            for (int q = 0; q < R; ++q)
                if (all[(size_t)p * R + q] != p * 100 + q + round) { fprintf(stderr, "rank %d: all-gather entry (%d,%d)\n", rank, p, q); return 1; }
        std::vector<double> wide(64), sum(64);
        for (int i = 0; i < 64; ++i) wide[i] = rank + 0.25 * i;
        CK(ncclAllReduce(wide.data(), sum.data(), 64, ncclFloat64, ncclSum, comm, st));
        for (int i = 0; i < 64; ++i)
            if (sum[i] != R * (R - 1) / 2.0 + 0.25 * i * R) { fprintf(stderr, "rank %d: all-reduce entry %d\n", rank, i); return 1; }
    }
    // an oversized collective is refused, not overrun
    std::vector<double> huge((1u << 16) / 8 + 8, 1.0);
    if (ncclAllReduce(huge.data(), huge.data(), huge.size(), ncclFloat64, ncclSum, comm, st) == ncclSuccess) { fprintf(stderr, "rank %d: oversized all-reduce accepted\n", rank); return 1; }
    CK(ncclCommDestroy(comm));
    return 0;
}

int main(int argc, char **argv)
{
    const int R = argc > 1 ? atoi(argv[1]) : 8, rounds = argc > 2 ? atoi(argv[2]) : 6;
    int rank = -1;
    ncclUniqueId id;
    CK(ncclGetUniqueId(&id));
    std::vector<pid_t> kids;
    for (int r = 0; r < R; ++r) {
        const pid_t p = fork();
        if (p < 0) { perror("fork"); return 2; }
        if (p == 0) _exit(run_rank(r, R, rounds, id));
        kids.push_back(p);
    }
    int bad = 0;
    for (pid_t p : kids) {
        int st = 0;
        if (waitpid(p, &st, 0) < 0 || !WIFEXITED(st) || WEXITSTATUS(st) != 0) ++bad;
    }
    printf("mock_rccl under ASan/UBSan: %d ranks x %d rounds, %d failed\n", R, rounds, bad);
    return bad ? 1 : 0;
}
