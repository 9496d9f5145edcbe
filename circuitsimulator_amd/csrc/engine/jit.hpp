// jit.hpp -- build side of the runtime specialisation: a private cache directory, a compiler
// child process started from an argv (no shell), and the checks a cached library must pass
// before it is loaded.
#pragma once

#include <string>
#include <vector>

namespace csim {

// Default cache directory: $CSIM_JIT_DIR, else $XDG_CACHE_HOME/csim_jit, else /tmp/csim_jit.<uid>.
std::string jitDefaultDir();

// Creates `dir` (mode 0700, parents of an explicit path as needed) and verifies that it is a real
// directory owned by the caller and not writable by group or others.  Returns "" or the reason.
std::string jitPrepareDir(const std::string& dir);

// A file this process may dlopen: a regular file (not a symlink) owned by the caller, not
// writable by group or others.
bool jitFileTrusted(const std::string& path);

// Runs argv[0] with argv (posix_spawn, no shell), stdout+stderr into logPath, and waits at most
// timeoutSec seconds (then SIGKILL).  Returns "" on exit status 0, else the reason.
std::string jitRun(const std::vector<std::string>& argv, const std::string& logPath, int timeoutSec);

} // namespace csim
