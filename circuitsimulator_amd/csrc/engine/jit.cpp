// jit.cpp -- see jit.hpp.
#include "jit.hpp"

#include <fcntl.h>
#include <signal.h>
#include <spawn.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

#include <cerrno>
#include <cstdlib>
#include <cstring>

extern char** environ;

namespace csim {

std::string jitDefaultDir()
{
    if (const char* d = std::getenv("CSIM_JIT_DIR")) if (*d) return d;
    if (const char* x = std::getenv("XDG_CACHE_HOME")) if (*x == '/') return std::string(x) + "/csim_jit";
    return "/tmp/csim_jit." + std::to_string(static_cast<long long>(geteuid()));
}

std::string jitPrepareDir(const std::string& dir)
{
    if (dir.empty() || dir[0] != '/') return "JIT directory must be an absolute path: '" + dir + "'";
    // parents first (default permissions), the leaf private
    for (std::size_t i = 1; i <= dir.size(); ++i) {
        if (i != dir.size() && dir[i] != '/') continue;
        const std::string part = dir.substr(0, i);
        const bool leaf = i == dir.size() || dir.find_first_not_of('/', i) == std::string::npos;
        if (mkdir(part.c_str(), leaf ? 0700 : 0755) != 0 && errno != EEXIST)
            return "cannot create " + part + ": " + std::strerror(errno);
        if (leaf) break;
    }
    struct stat st;
    if (lstat(dir.c_str(), &st) != 0) return "cannot stat " + dir + ": " + std::strerror(errno);
    if (!S_ISDIR(st.st_mode)) return dir + " is not a directory (a symlink is refused)";
    if (st.st_uid != geteuid()) return dir + " is owned by another user";
    if (st.st_mode & (S_IWGRP | S_IWOTH)) return dir + " is writable by group or others";
    return std::string();
}

bool jitFileTrusted(const std::string& path)
{
    struct stat st;
    if (lstat(path.c_str(), &st) != 0) return false;
    return S_ISREG(st.st_mode) && st.st_uid == geteuid() && !(st.st_mode & (S_IWGRP | S_IWOTH));
}

std::string jitRun(const std::vector<std::string>& argv, const std::string& logPath, int timeoutSec)
{
    if (argv.empty()) return "empty command";
    std::vector<char*> av;
    for (const std::string& a : argv) av.push_back(const_cast<char*>(a.c_str()));
    av.push_back(nullptr);
    posix_spawn_file_actions_t fa;
    posix_spawn_file_actions_init(&fa);
    posix_spawn_file_actions_addopen(&fa, 1, logPath.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0600);
    posix_spawn_file_actions_adddup2(&fa, 1, 2);
    posix_spawn_file_actions_addopen(&fa, 0, "/dev/null", O_RDONLY, 0);
    posix_spawnattr_t at;
    posix_spawnattr_init(&at);
    posix_spawnattr_setflags(&at, POSIX_SPAWN_SETPGROUP);      // own process group: the timeout kills the compiler's children too
    posix_spawnattr_setpgroup(&at, 0);
    // The compiler must not inherit a profiler's or tool's preload: under rocprofv3 the preloaded library has
    // initialised the GPU in this process, and every exec of the hipcc -> clang -> lld chain with that
    // environment would be an exec from a GPU-initialised process image (forbidden on some pools).
    std::vector<char*> envp;
    for (char** e = environ; e && *e; ++e) {
        static const char* const drop[] = {"LD_PRELOAD=", "ROCP_TOOL_LIBRARIES=", "ROCPROFILER_", "ROCPROF_", "ROCP_",
                                           "HSA_TOOLS_LIB=", "HSA_TOOLS_REPORT_LOAD_FAILURE=", "ROCTRACER_", "OMPT_TOOL_LIBRARIES="};
        bool keep = true;
        for (const char* d : drop) keep = keep && std::strncmp(*e, d, std::strlen(d)) != 0;
        if (keep) envp.push_back(*e);
    }
    envp.push_back(nullptr);
    pid_t pid = 0;
    const int rc = posix_spawn(&pid, av[0], &fa, &at, av.data(), envp.data());
    posix_spawn_file_actions_destroy(&fa);
    posix_spawnattr_destroy(&at);
    if (rc != 0) return std::string("cannot start ") + argv[0] + ": " + std::strerror(rc);
    struct timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (;;) {
        int status = 0;
        const pid_t w = waitpid(pid, &status, WNOHANG);
        if (w == pid) {
            if (WIFEXITED(status) && WEXITSTATUS(status) == 0) return std::string();
            return argv[0] + (WIFEXITED(status) ? " exited with status " + std::to_string(WEXITSTATUS(status))
                                                : std::string(" was killed by a signal")) + ", see " + logPath;
        }
        if (w < 0 && errno != EINTR) return std::string("waitpid: ") + std::strerror(errno);
        struct timespec now;
        clock_gettime(CLOCK_MONOTONIC, &now);
        if (now.tv_sec - t0.tv_sec > timeoutSec) {
            kill(-pid, SIGKILL);
            kill(pid, SIGKILL);
            (void)waitpid(pid, &status, 0);
            return argv[0] + " did not finish within " + std::to_string(timeoutSec) + " s";
        }
        struct timespec nap = {0, 20 * 1000 * 1000};
        nanosleep(&nap, nullptr);
    }
}

} // namespace csim
