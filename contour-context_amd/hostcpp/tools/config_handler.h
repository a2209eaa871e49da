// `yamlLoader` with the reference's interface (include/tools/config_handler.h:13-85), without OpenCV's FileStorage:
// a reader for the subset of YAML the shipped configs use (config/batch_bin_test_config.yaml): block maps nested by
// indentation, scalars, quoted strings, one-line flow sequences `[a, b, c]`, `#` comments, the `%YAML:1.0` / `---`
// preamble.  A missing key leaves the destination untouched and says so, as the reference does.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

struct yamlLoader {
  std::map<std::string, std::string> flat;  // "a/b/c" -> raw scalar or "[..]" text
  bool opened = false;

  explicit yamlLoader(const std::string &config_fpath) { read(config_fpath); }

  void read(const std::string &config_fpath) {
    flat.clear();
    opened = false;
    std::ifstream f(config_fpath);
    if (!f.good()) return;
    opened = true;
    std::vector<std::pair<int, std::string>> stack;  // (indent, key) of the open maps
    std::string line;
    while (std::getline(f, line)) {
      const std::string body = strip_comment(line);
      size_t ind = 0;
      while (ind < body.size() && body[ind] == ' ') ind++;
      if (ind == body.size()) continue;
      if (body[ind] == '%' || body.compare(ind, 3, "---") == 0) continue;
      const size_t colon = find_colon(body, ind);
      if (colon == std::string::npos) continue;
      const std::string key = trim(body.substr(ind, colon - ind));
      const std::string val = trim(body.substr(colon + 1));
      while (!stack.empty() && stack.back().first >= (int)ind) stack.pop_back();
      std::string path;
      for (const auto &s : stack) path += s.second + "/";
      path += key;
      if (val.empty())
        stack.emplace_back((int)ind, key);
      else
        flat[path] = val;
    }
  }

  void close() {
    flat.clear();
    opened = false;
  }

  template <typename T>
  void loadOneConfig(const std::vector<std::string> &keys, T &container) const {
    const std::string *v = lookup(keys);
    if (!v) return;
    convert(unquote(*v), container);
    std::cout << ": " << container << std::endl;
  }

  template <typename T>
  void loadSeqConfig(const std::vector<std::string> &keys, std::vector<T> &container) const {
    const std::string *v = lookup(keys);
    if (!v) return;
    if (v->size() < 2 || v->front() != '[' || v->back() != ']') {
      fprintf(stderr, "Check failed: node is a sequence\n");
      abort();
    }
    container.clear();
    std::stringstream ss(v->substr(1, v->size() - 2));
    std::string item;
    while (std::getline(ss, item, ',')) {
      item = trim(item);
      if (item.empty()) continue;
      T x;
      convert(unquote(item), x);
      container.emplace_back(x);
    }
    std::cout << ": ";
    for (const auto &d : container) std::cout << d << ", ";
    std::cout << std::endl;
  }

 private:
  const std::string *lookup(const std::vector<std::string> &keys) const {
    if (keys.empty()) {
      fprintf(stderr, "Check failed: !keys.empty()\n");
      abort();
    }
    std::string path;
    for (size_t i = 0; i < keys.size(); i++) {
      path += (i ? "/" : "") + keys[i];
      std::cout << "\"" << keys[i] << "\"->";
    }
    auto it = flat.find(path);
    if (it == flat.end()) {
      printf(": [!] Cannot find the specified config parameter!\n");
      return nullptr;
    }
    return &it->second;
  }
  static std::string trim(const std::string &s) {
    size_t a = 0, b = s.size();
    while (a < b && (s[a] == ' ' || s[a] == '\t' || s[a] == '\r')) a++;
    while (b > a && (s[b - 1] == ' ' || s[b - 1] == '\t' || s[b - 1] == '\r')) b--;
    return s.substr(a, b - a);
  }
  static std::string unquote(const std::string &s) {
    if (s.size() >= 2 && (s.front() == '"' || s.front() == '\'') && s.back() == s.front()) return s.substr(1, s.size() - 2);
    return s;
  }
  static std::string strip_comment(const std::string &s) {  // '#' outside quotes starts a comment
    char q = 0;
    for (size_t i = 0; i < s.size(); i++) {
      if (q) {
        if (s[i] == q) q = 0;
      } else if (s[i] == '"' || s[i] == '\'') {
        q = s[i];
      } else if (s[i] == '#') {
        return s.substr(0, i);
      }
    }
    return s;
  }
  static size_t find_colon(const std::string &s, size_t from) {  // first ':' outside quotes
    char q = 0;
    for (size_t i = from; i < s.size(); i++) {
      if (q) {
        if (s[i] == q) q = 0;
      } else if (s[i] == '"' || s[i] == '\'') {
        q = s[i];
      } else if (s[i] == ':') {
        return i;
      }
    }
    return std::string::npos;
  }
  static void convert(const std::string &s, std::string &o) { o = s; }
  static void convert(const std::string &s, int &o) { o = (int)std::strtod(s.c_str(), nullptr); }
  static void convert(const std::string &s, float &o) { o = std::strtof(s.c_str(), nullptr); }
  static void convert(const std::string &s, double &o) { o = std::strtod(s.c_str(), nullptr); }
};
