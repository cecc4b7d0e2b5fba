// File rendezvous for the RCCL unique id: how a Fortran (or any non-torch) job of N processes,
// started by whatever launcher exported RANK/WORLD_SIZE, hands rank 0's ncclUniqueId to the
// others without MPI.  It replaces what MPI_Init gives the reference for free
// (parallel/parallel_utils_mod.f90:77-90).  Host code only: no HIP call, no child process
// (nothing may fork/exec once the GPU is initialised), testable on a machine without a GPU.
//
// Record (256 bytes, written to <path>.tmp.<pid> and published with rename(2), so a reader sees
// all of it or none):
//     0  "DLESMRV1"
//     8  int64   start time of the publishing process, seconds since the epoch
//    16  char[112] job token, NUL padded  ("<world size>:<launcher run id>")
//   128  char[128] ncclUniqueId
// A file left behind by a dead job is rejected twice over: its token differs when the launcher
// provides a run id, and its publisher did not start within `slack` seconds of the reader (the
// ranks of one job start together; DLESM_RENDEZVOUS_SLACK_S, default 120).  Rank 0 also removes
// whatever it finds under the name before it creates the new id.
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <string>

#include "dlesm_error.h"
#include "dlesm_hip.h"

using dlesm::fail;

#define DLESM_REQUIRE(cond, ...)                                                            \
    do {                                                                                    \
        if (!(cond)) return ::dlesm::fail(DLESM_EINVAL, __VA_ARGS__);                       \
    } while (0)

namespace {

constexpr char MAGIC[8] = {'D', 'L', 'E', 'S', 'M', 'R', 'V', '1'};
constexpr int TOKEN_BYTES = 112, RECORD_BYTES = 256;

// start of this process in epoch seconds: btime (/proc/stat) + starttime ticks (/proc/self/stat
// field 22); the time the library was loaded if /proc cannot be read
long long process_start()
{
    static long long cached = 0;
    if (cached) return cached;
    long long btime = 0, ticks = -1;
    if (FILE *f = fopen("/proc/stat", "r")) {
        char line[256];
        while (fgets(line, sizeof line, f))
            if (sscanf(line, "btime %lld", &btime) == 1) break;
        fclose(f);
    }
    if (FILE *f = fopen("/proc/self/stat", "r")) {
        char buf[2048];
        size_t n = fread(buf, 1, sizeof buf - 1, f);
        fclose(f);
        buf[n] = 0;
        if (const char *p = strrchr(buf, ')')) { // the command name may contain spaces: fields restart after it
            int field = 2;                       // p points at the end of field 2 (comm)
            for (const char *c = p + 1; *c; ) {
                while (*c == ' ') c++;
                if (!*c) break;
                if (++field == 22) { ticks = atoll(c); break; }
                while (*c && *c != ' ') c++;
            }
        }
    }
    const long hz = sysconf(_SC_CLK_TCK);
    cached = (btime > 0 && ticks >= 0 && hz > 0) ? btime + ticks / hz : (long long)time(nullptr);
    return cached;
}

long long slack_seconds()
{
    if (const char *e = getenv("DLESM_RENDEZVOUS_SLACK_S")) {
        const long long v = atoll(e);
        if (v > 0) return v;
    }
    return 120;
}

} // namespace

extern "C" int dlesm_rendezvous_remove(const char *path)
{
    DLESM_REQUIRE(path != nullptr && path[0], "rendezvous: empty path");
    if (unlink(path) != 0 && errno != ENOENT)
        return fail(DLESM_EINVAL, "rendezvous: cannot remove %s: %s", path, strerror(errno));
    return DLESM_OK;
}

extern "C" int dlesm_rendezvous_publish(const char *path, const void *id, const char *token)
{
    DLESM_REQUIRE(path != nullptr && path[0] && id != nullptr && token != nullptr, "rendezvous: bad arguments");
    DLESM_REQUIRE(strlen(token) < (size_t)TOKEN_BYTES, "rendezvous: token longer than %d bytes", TOKEN_BYTES - 1);
    char rec[RECORD_BYTES];
    memset(rec, 0, sizeof rec);
    memcpy(rec, MAGIC, 8);
    const long long t0 = process_start();
    memcpy(rec + 8, &t0, 8);
    strncpy(rec + 16, token, TOKEN_BYTES - 1);
    memcpy(rec + 128, id, DLESM_UNIQUE_ID_BYTES);
    char tmp[4096];
    const int n = snprintf(tmp, sizeof tmp, "%s.tmp.%ld", path, (long)getpid());
    DLESM_REQUIRE(n > 0 && (size_t)n < sizeof tmp, "rendezvous: path too long");
    const int fd = open(tmp, O_WRONLY | O_CREAT | O_TRUNC, 0600);
    if (fd < 0) return fail(DLESM_EINVAL, "rendezvous: cannot create %s: %s", tmp, strerror(errno));
    const ssize_t w = write(fd, rec, sizeof rec);
    const int sync_rc = fsync(fd), close_rc = close(fd);
    if (w != (ssize_t)sizeof rec || sync_rc != 0 || close_rc != 0) {
        unlink(tmp);
        return fail(DLESM_EINVAL, "rendezvous: short write to %s: %s", tmp, strerror(errno));
    }
    if (rename(tmp, path) != 0) {
        const int e = errno;
        unlink(tmp);
        return fail(DLESM_EINVAL, "rendezvous: rename %s -> %s: %s", tmp, path, strerror(e));
    }
    return DLESM_OK;
}

extern "C" int dlesm_rendezvous_fetch(const char *path, void *id, const char *token, int timeout_ms)
{
    DLESM_REQUIRE(path != nullptr && path[0] && id != nullptr && token != nullptr, "rendezvous: bad arguments");
    const long long mine = process_start(), slack = slack_seconds();
    char why[256] = "no file appeared";
    struct timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (;;) {
        const int fd = open(path, O_RDONLY);
        if (fd >= 0) {
            char rec[RECORD_BYTES];
            const ssize_t r = read(fd, rec, sizeof rec);
            close(fd);
            long long theirs = 0;
            if (r == (ssize_t)sizeof rec && memcmp(rec, MAGIC, 8) == 0) {
                memcpy(&theirs, rec + 8, 8);
                rec[16 + TOKEN_BYTES - 1] = 0;
                const long long apart = theirs > mine ? theirs - mine : mine - theirs;
                if (strcmp(rec + 16, token) != 0)
                    snprintf(why, sizeof why, "stale file: job token '%.60s', this job is '%.60s'", rec + 16, token);
                else if (apart > slack)
                    snprintf(why, sizeof why, "stale file: its publisher started %lld s apart from this process "
                             "(limit %lld, DLESM_RENDEZVOUS_SLACK_S)", apart, slack);
                else {
                    memcpy(id, rec + 128, DLESM_UNIQUE_ID_BYTES);
                    return DLESM_OK;
                }
            } else {
                snprintf(why, sizeof why, "not a rendezvous record (%zd bytes)", r);
            }
        }
        struct timespec now;
        clock_gettime(CLOCK_MONOTONIC, &now);
        const long long waited = (now.tv_sec - t0.tv_sec) * 1000LL + (now.tv_nsec - t0.tv_nsec) / 1000000LL;
        if (waited >= timeout_ms) break;
        const struct timespec nap = {0, 10 * 1000 * 1000}; // 10 ms, sleeping (no busy wait)
        nanosleep(&nap, nullptr);
    }
    return fail(DLESM_EINVAL, "rendezvous: no usable RCCL id at %s after %d ms (%s)", path, timeout_ms, why);
}

// Dry runs (DLESM_DRY_COMMS=2) create no communicator, so nothing holds rank 0 back until the
// others have read the record (with RCCL, ncclCommInitRank does).  Readers acknowledge with an
// empty file <path>.ack.<rank>; rank 0 waits for all of them and removes them.
extern "C" int dlesm_rendezvous_ack(const char *path, int rank0)
{
    DLESM_REQUIRE(path != nullptr && path[0] && rank0 > 0, "rendezvous: bad arguments");
    char name[4096];
    const int n = snprintf(name, sizeof name, "%s.ack.%d", path, rank0);
    DLESM_REQUIRE(n > 0 && (size_t)n < sizeof name, "rendezvous: path too long");
    const int fd = open(name, O_WRONLY | O_CREAT | O_TRUNC, 0600);
    if (fd < 0) return fail(DLESM_EINVAL, "rendezvous: cannot create %s: %s", name, strerror(errno));
    close(fd);
    return DLESM_OK;
}

extern "C" int dlesm_rendezvous_wait_acks(const char *path, int nranks, int timeout_ms)
{
    DLESM_REQUIRE(path != nullptr && path[0] && nranks >= 1, "rendezvous: bad arguments");
    struct timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int r = 1; r < nranks; r++) {
        char name[4096];
        snprintf(name, sizeof name, "%s.ack.%d", path, r);
        for (;;) {
            if (unlink(name) == 0) break;
            struct timespec now;
            clock_gettime(CLOCK_MONOTONIC, &now);
            const long long waited = (now.tv_sec - t0.tv_sec) * 1000LL + (now.tv_nsec - t0.tv_nsec) / 1000000LL;
            if (waited >= timeout_ms)
                return fail(DLESM_EINVAL, "rendezvous: rank %d did not pick up the id within %d ms", r, timeout_ms);
            const struct timespec nap = {0, 10 * 1000 * 1000};
            nanosleep(&nap, nullptr);
        }
    }
    return DLESM_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// The BOARD: a host-side all-gather between the processes of one job through files in /dev/shm -- the control plane of
// the mailbox transport when there is no RCCL communicator (dlesm_comm_init_mailbox): mailbox descriptors at plan
// creation, the eight bytes of a global sum, the hand-shake of a gather.  What MPI_Allgather on a few bytes is to the
// reference.  Operation k of rank r is the file <prefix>.<k>.<r>, written whole and published with rename(2); a rank
// has finished operation k when it has read the n files of k.  A rank writes its file of k+1 only after that, so once
// a rank has read all files of k+1 nobody needs its file of k any more: it removes it then.  Host code only.
namespace {

struct Board {
    std::string prefix;
    int rank = -1, n = 0;
    long op = 0;
} g_board;

int board_write(const std::string &name, const void *data, size_t bytes)
{
    // /dev/shm is world-writable and the names are predictable: the temporary file must be a NEW regular file of ours --
    // O_EXCL | O_NOFOLLOW refuse a pre-created file and a planted symbolic link alike (a leftover of a killed run of this
    // very process name is removed first; rename(2) then replaces the final name atomically, whatever sits there)
    const std::string tmp = name + ".tmp." + std::to_string((long)getpid());
    (void)unlink(tmp.c_str());
    const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
    if (fd < 0) return fail(DLESM_EINVAL, "board: cannot create %s: %s", tmp.c_str(), strerror(errno));
    size_t done = 0;
    while (done < bytes) {
        const ssize_t w = write(fd, (const char *)data + done, bytes - done);
        if (w <= 0) { close(fd); unlink(tmp.c_str()); return fail(DLESM_EINVAL, "board: write to %s: %s", tmp.c_str(), strerror(errno)); }
        done += (size_t)w;
    }
    if (close(fd) != 0 || rename(tmp.c_str(), name.c_str()) != 0) {
        unlink(tmp.c_str());
        return fail(DLESM_EINVAL, "board: cannot publish %s: %s", name.c_str(), strerror(errno));
    }
    return DLESM_OK;
}

// A rank that stops (parallel_abort -> dlesm_board_abort) leaves a note; the others, waiting for a file of that rank that
// will never come, find the note and fail with its text instead of sitting out the time-out -- what MPI_Abort does for
// the reference (parallel_utils_mod.f90:104-111), as far as a host-side board can.
int board_aborted(char *why, size_t n)
{
    const std::string note = g_board.prefix + ".abort";
    const int fd = open(note.c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
    if (fd < 0) return 0;
    const ssize_t r = read(fd, why, n - 1);
    close(fd);
    why[r > 0 ? r : 0] = 0;
    return 1;
}

int board_read(const std::string &name, void *data, size_t bytes, int timeout_ms)
{
    struct timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (long spin = 0;; spin++) {
        if (spin % 64 == 63) {
            char why[400];
            if (board_aborted(why, sizeof why)) return fail(DLESM_EABORT, "board: another rank stopped the job: %s", why);
        }
        const int fd = open(name.c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
        if (fd >= 0) {
            size_t done = 0;
            while (done < bytes) {
                const ssize_t r = read(fd, (char *)data + done, bytes - done);
                if (r <= 0) break;
                done += (size_t)r;
            }
            close(fd);
            if (done == bytes) return DLESM_OK;
            return fail(DLESM_EINVAL, "board: %s holds %zu bytes, %zu expected (ranks out of step?)", name.c_str(), done, bytes);
        }
        struct timespec now;
        clock_gettime(CLOCK_MONOTONIC, &now);
        const long long waited = (now.tv_sec - t0.tv_sec) * 1000LL + (now.tv_nsec - t0.tv_nsec) / 1000000LL;
        if (waited >= timeout_ms) return fail(DLESM_EINVAL, "board: %s did not appear within %d ms", name.c_str(), timeout_ms);
        const struct timespec nap = {0, 200 * 1000};     // 0.2 ms, sleeping
        nanosleep(&nap, nullptr);
    }
}

std::string board_name(long op, int r) { return g_board.prefix + "." + std::to_string(op) + "." + std::to_string(r); }

int board_timeout_ms()
{
    if (const char *e = getenv("DLESM_BOARD_TIMEOUT_S")) {
        const long v = atol(e);
        if (v > 0) return (int)(v * 1000);
    }
    return 600 * 1000;      // a neighbour may be seconds or minutes behind (it waits in MPI without limit in the reference)
}

} // namespace

// 128 bytes (the size of an RCCL unique id, so that it travels the same way): a name no other job has
extern "C" int dlesm_board_nonce(void *id)
{
    DLESM_REQUIRE(id != nullptr, "null id buffer");
    char buf[DLESM_UNIQUE_ID_BYTES];
    memset(buf, 0, sizeof buf);
    struct timespec now;
    clock_gettime(CLOCK_REALTIME, &now);
    snprintf(buf, sizeof buf, "mbx-%lld-%ld-%ld", (long long)now.tv_sec, (long)now.tv_nsec, (long)getpid());
    memcpy(id, buf, sizeof buf);
    return DLESM_OK;
}

extern "C" int dlesm_board_open(const void *id, int nranks, int rank0)
{
    DLESM_REQUIRE(id != nullptr && nranks >= 1 && rank0 >= 0 && rank0 < nranks, "board: rank %d of %d", rank0, nranks);
    DLESM_REQUIRE(g_board.rank < 0, "board: already open");
    char name[DLESM_UNIQUE_ID_BYTES + 1];
    memcpy(name, id, DLESM_UNIQUE_ID_BYTES);
    name[DLESM_UNIQUE_ID_BYTES] = 0;
    for (const char *c = name; *c; c++)
        DLESM_REQUIRE((*c >= '0' && *c <= '9') || (*c >= 'a' && *c <= 'z') || (*c >= 'A' && *c <= 'Z') || *c == '-' || *c == '_',
                      "board: the session name may hold letters, digits, '-' and '_' only");
    DLESM_REQUIRE(name[0], "board: empty session name");
    const char *dir = getenv("DLESM_BOARD_DIR");
    g_board.prefix = std::string(dir && *dir ? dir : "/dev/shm") + "/dlesm_board_" + name;
    g_board.rank = rank0;
    g_board.n = nranks;
    g_board.op = 0;
    return DLESM_OK;
}

extern "C" int dlesm_board_is_open(void) { return g_board.rank >= 0 ? 1 : 0; }

// every rank contributes `bytes`; `all` (nranks x bytes, rank order) may be null on ranks that only contribute
// (a gather to a root: the others still take part, so that the operation counts stay in step)
extern "C" int dlesm_board_allgather(const void *mine, size_t bytes, void *all)
{
    DLESM_REQUIRE(g_board.rank >= 0, "board: not open (dlesm_comm_init_mailbox)");
    DLESM_REQUIRE(mine != nullptr || bytes == 0, "board: null contribution");
    const long op = ++g_board.op;
    const int to = board_timeout_ms();
    if (int rc = board_write(board_name(op, g_board.rank), mine, bytes)) return rc;
    std::string scratch;
    for (int r = 0; r < g_board.n; r++) {
        void *dst = all ? (char *)all + (size_t)r * bytes : nullptr;
        if (!dst) { scratch.resize(bytes); dst = &scratch[0]; }
        if (int rc = board_read(board_name(op, r), dst, bytes, to)) return rc;
    }
    if (op > 1) unlink(board_name(op - 1, g_board.rank).c_str());
    return DLESM_OK;
}

extern "C" int dlesm_board_abort(const char *msg)
{
    if (g_board.rank < 0) return DLESM_OK;
    char text[400];
    snprintf(text, sizeof text, "rank %d: %.300s", g_board.rank, msg ? msg : "");
    (void)board_write(g_board.prefix + ".abort", text, strlen(text));
    // this rank is leaving: its outstanding operation files are of no use to anybody (the others fail on the note).  The
    // note itself has to outlive this process -- the waiting ranks read it after we are gone; the LAST of them to fail on it
    // cannot be told apart from the first, so the note stays (a few hundred bytes under a session name no other job will
    // ever use, dlesm_board_nonce) unless a rank of the job still gets to dlesm_board_close, which removes it.
    for (long op = g_board.op > 1 ? g_board.op - 1 : 1; op <= g_board.op; op++) unlink(board_name(op, g_board.rank).c_str());
    g_board = Board{};
    return DLESM_OK;
}

extern "C" int dlesm_board_close(void)
{
    if (g_board.rank < 0) return DLESM_OK;
    char c = 0;
    int rc = dlesm_board_allgather(&c, 1, nullptr);                 // A: after it, nobody needs my files of earlier operations
    const long a = g_board.op;
    if (!rc) rc = dlesm_board_allgather(&c, 1, nullptr);            // B: after it, nobody needs my file of A (removed by B itself)
    const long b = g_board.op;
    if (!rc) {
        // the files of B: every rank but 0 says when it has read them all; rank 0 then removes them and the notes
        if (g_board.rank > 0) {
            rc = board_write(g_board.prefix + ".done." + std::to_string(g_board.rank), &c, 1);
        } else {
            for (int r = 1; r < g_board.n && !rc; r++) {
                const std::string note = g_board.prefix + ".done." + std::to_string(r);
                rc = board_read(note, &c, 1, board_timeout_ms());
                unlink(note.c_str());
            }
            for (int r = 0; r < g_board.n; r++) unlink(board_name(b, r).c_str());
        }
    }
    (void)a;
    if (rc) {      // a failed close (a rank stopped, a time-out): nothing of THIS rank stays behind all the same
        for (long op = g_board.op > 2 ? g_board.op - 2 : 1; op <= g_board.op; op++) unlink(board_name(op, g_board.rank).c_str());
        unlink((g_board.prefix + ".done." + std::to_string(g_board.rank)).c_str());
        if (g_board.rank == 0) unlink((g_board.prefix + ".abort").c_str());
    }
    g_board = Board{};
    return rc;
}
