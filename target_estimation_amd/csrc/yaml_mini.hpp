// yaml_mini.hpp -- dependency-free reader for the reference's model files
// (models/model_*_params.yaml: `type`, `frequency`, and flat row-major flow sequences `Q`, `R`,
// `P`; reference loader: src/target_manager.cpp:18-104, which uses yaml-cpp).
#pragma once
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

namespace te {

struct ModelFile {
  std::string type;
  double frequency = 0.0;
  bool has_frequency = false;
  std::map<std::string, std::vector<double>> seqs;
};

inline std::string yaml_trim(const std::string& s) {
  size_t b = s.find_first_not_of(" \t\r\n");
  if (b == std::string::npos) return "";
  size_t e = s.find_last_not_of(" \t\r\n");
  return s.substr(b, e - b + 1);
}

// Returns false if the file cannot be opened or a sequence is malformed.
inline bool load_model_file(const std::string& path, ModelFile& out, std::string& err) {
  std::ifstream f(path.c_str());
  if (!f.is_open()) { err = "bad file: " + path; return false; }
  std::stringstream ss;
  ss << f.rdbuf();
  const std::string text = ss.str();
  size_t pos = 0;
  while (pos < text.size()) {
    size_t eol = text.find('\n', pos);
    if (eol == std::string::npos) eol = text.size();
    std::string line = text.substr(pos, eol - pos);
    size_t hash = line.find('#');
    if (hash != std::string::npos) line = line.substr(0, hash);
    size_t colon = line.find(':');
    if (colon == std::string::npos) { pos = eol + 1; continue; }
    const std::string key = yaml_trim(line.substr(0, colon));
    std::string val = yaml_trim(line.substr(colon + 1));
    size_t next = eol + 1;
    if (!val.empty() && val[0] == '[') {
      // flow sequence, possibly spanning several lines
      size_t start = text.find('[', pos + colon);
      size_t close = text.find(']', start);
      if (close == std::string::npos) { err = "unterminated sequence for key " + key; return false; }
      std::string body = text.substr(start + 1, close - start - 1);
      std::vector<double> v;
      const char* p = body.c_str();
      while (*p) {
        while (*p == ' ' || *p == ',' || *p == '\n' || *p == '\t' || *p == '\r') ++p;
        if (!*p) break;
        char* endp = nullptr;
        double d = std::strtod(p, &endp);
        if (endp == p) { err = "bad number in sequence " + key; return false; }
        v.push_back(d);
        p = endp;
      }
      out.seqs[key] = v;
      next = text.find('\n', close);
      next = (next == std::string::npos) ? text.size() : next + 1;
    } else if (key == "type") {
      if (val.size() >= 2 && (val[0] == '"' || val[0] == '\'')) val = val.substr(1, val.size() - 2);
      out.type = val;
    } else if (key == "frequency") {
      out.frequency = std::strtod(val.c_str(), nullptr);
      out.has_frequency = true;
    }
    pos = next;
  }
  return true;
}

}  // namespace te
