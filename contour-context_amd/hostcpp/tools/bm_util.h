// Mirror of the reference's tools/bm_util.h interface (TicToc, SequentialTimeProfiler): the stage timers its drivers and
// ContourDB::queryRangedKNN record ("make bev", "KNN search", "Constell", "L2 opt", "Update database") and the table
// they print, so that timings of this build line up with log/timing_cont2_paper.txt.  Pure STL.
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <ctime>
#include <map>
#include <string>
#include <utility>
#include <vector>

class TicToc {
public:
  TicToc() { tic(); }
  void tic() { t0_ = std::chrono::steady_clock::now(); }
  double toc() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0_).count(); }
  double toctic() {
    const double r = toc();
    tic();
    return r;
  }

private:
  std::chrono::steady_clock::time_point t0_;
};

class SequentialTimeProfiler {
  struct Entry {
    int idx = 0, cnt = 0;
    double sum = 0, sum_sq = 0;
  };
  TicToc clk_;
  std::map<std::string, Entry> logs_;
  int loops_ = 0;
  std::string desc_;

  void add(const std::string &name, double dt) {
    auto it = logs_.find(name);
    if (it == logs_.end()) {
      Entry e;
      e.idx = (int)logs_.size();
      it = logs_.emplace(name, e).first;
    }
    it->second.cnt++;
    it->second.sum += dt;
    it->second.sum_sq += dt * dt;
  }
  void print(std::FILE *fp, bool sort_by_cost) const {
    size_t w = 5;
    std::vector<std::pair<std::string, Entry>> v(logs_.begin(), logs_.end());
    for (const auto &e : v) w = std::max(w, e.first.size());
    std::sort(v.begin(), v.end(), [&](const std::pair<std::string, Entry> &a, const std::pair<std::string, Entry> &b) {
      return sort_by_cost ? a.second.sum > b.second.sum : a.second.idx < b.second.idx;
    });
    double total = 0, accum = 0;
    for (const auto &e : v) total += e.second.sum;
    std::fprintf(fp, "\n=== Time Profiling @%s ===\n=== Description: %s\n", getTimeString().c_str(), desc_.c_str());
    std::fprintf(fp, "%5s %*s %10s %10s %10s %10s %10s %10s\n", "Index", (int)w, "Name", "Count", "Average", "Stddev", "Per loop", "Loop %",
                 "Accum %");
    for (const auto &e : v) {
      const Entry &g = e.second;
      const double mean = g.sum / g.cnt;
      const double sd = g.cnt > 1 ? std::sqrt(std::max(0.0, (g.sum_sq - g.cnt * mean * mean) / (g.cnt - 1))) : 0.0;
      accum += g.sum;
      std::fprintf(fp, "%5d %*s %10d %10.2e %10.2e %10.2e %10.2f %10.2f\n", g.idx, (int)w, e.first.c_str(), g.cnt, mean, sd,
                   loops_ > 0 ? g.sum / loops_ : 0.0, total > 0 ? g.sum / total * 100 : 0.0, total > 0 ? accum / total * 100 : 0.0);
    }
    std::fprintf(fp, "%5s %*s %10d %10s %10s %10.2e %10s %10s\n", "*", (int)w, "*sum", loops_, "*", "*", loops_ > 0 ? total / loops_ : 0.0, "*",
                 "*");
  }

public:
  SequentialTimeProfiler() = default;
  SequentialTimeProfiler(const std::string &name) : desc_(name) {}
  std::string getDesc() const { return desc_; }
  static std::string getTimeString() {
    std::time_t now = std::time(nullptr);
    char buf[80];
    std::strftime(buf, sizeof(buf), "%Y-%m-%d %a %X %z", std::localtime(&now));
    return buf;
  }
  void start() { clk_.tic(); }
  void record(const std::string &name) {  // record and restart: sequential stages
    add(name, clk_.toc());
    clk_.tic();
  }
  void record(const std::string &name, double &dt_curr) {
    dt_curr = clk_.toc();
    add(name, dt_curr);
    clk_.tic();
  }
  void lap() { loops_++; }
  void addSample(const std::string &name, double seconds) { add(name, seconds); }  // extension: a duration measured elsewhere
  double total(const std::string &name) const {  // extension: accumulated seconds of a stage (0 if never recorded)
    auto it = logs_.find(name);
    return it == logs_.end() ? 0.0 : it->second.sum;
  }
  void printScreen(bool sort_by_cost = false) const { print(stdout, sort_by_cost); }
  void printFile(const std::string &fpath, bool sort_by_cost = false) const {
    if (std::FILE *fp = std::fopen(fpath.c_str(), "a")) {
      print(fp, sort_by_cost);
      std::fclose(fp);
    }
  }
};
