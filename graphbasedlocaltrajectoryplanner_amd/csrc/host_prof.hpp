// Host wall-clock accounting of the latency path (diagnostics only): LTPL_HOST_PROF=1 accumulates the time spent in named
// sections of the C-ABI entry points and of the planner state machine and prints mean microseconds per call at process exit.
// One branch per section when disabled.
#pragma once
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace ltplprof {
struct Slot { const char* name; double us; long calls; };
struct Table {
    bool on; int n; Slot s[64];
    Table() : on(false), n(0) { const char* e = getenv("LTPL_HOST_PROF"); on = e && atoi(e) != 0; }
    ~Table()
    {
        if (!on) return;
        for (int i = 0; i < n; ++i)
            fprintf(stderr, "[ltpl host prof] %-28s calls %7ld  mean %8.2f us\n", s[i].name, s[i].calls, s[i].calls ? s[i].us / (double)s[i].calls : 0.0);
    }
    Slot* slot(const char* name)
    {
        for (int i = 0; i < n; ++i) if (s[i].name == name || !strcmp(s[i].name, name)) return &s[i];
        if (n >= 64) return &s[63];
        s[n].name = name; s[n].us = 0.0; s[n].calls = 0;
        return &s[n++];
    }
};
inline Table& table() { static Table t; return t; }
struct Scope {
    Slot* sl; std::chrono::steady_clock::time_point t0;
    explicit Scope(const char* name) : sl(nullptr) { Table& t = table(); if (t.on) { sl = t.slot(name); t0 = std::chrono::steady_clock::now(); } }
    void stop()
    {
        if (!sl) return;
        sl->us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); sl->calls += 1; sl = nullptr;
    }
    ~Scope() { stop(); }
};
}
#define LTPL_PROF(var, name) ltplprof::Scope var(name)
