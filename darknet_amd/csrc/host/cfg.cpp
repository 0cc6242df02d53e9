// cfg.cpp -- .cfg reader and typed option lookup.
// Behaviour follows the reference (src/parser.cpp:59-100 ReadSections,
// src/utils.cpp:133-148 strip, src/option_list.cpp:134-243): every space, tab,
// CR and LF is removed from a line; '[' opens a section; lines starting with
// '#', ';' or empty are skipped; everything else is key=value split at the first
// '='; non-quiet getters print the default they fall back to; options never
// looked up are reported as "Unused field".
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dk_host.h"

void* xcalloc(size_t nmemb, size_t size)
{
  void* p = calloc(nmemb ? nmemb : 1, size ? size : 1);
  if (!p)
  {
    fprintf(stderr, "calloc failed (%zu x %zu)\n", nmemb, size);
    exit(EXIT_FAILURE);
  }
  return p;
}

void* xrealloc(void* ptr, size_t size)
{
  void* p = realloc(ptr, size ? size : 1);
  if (!p)
  {
    fprintf(stderr, "realloc failed (%zu)\n", size);
    exit(EXIT_FAILURE);
  }
  return p;
}

void error(const char* s)
{
  // src/utils.cpp:120-125
  perror(s);
  exit(EXIT_FAILURE);
}

void FileError(const char* s)
{
  fprintf(stderr, "Couldn't open file: %s\n", s);
  exit(EXIT_FAILURE);
}

bool ReadSections(const char* filename, std::vector<Section>& out)
{
  FILE* f = fopen(filename, "r");
  if (!f)
    return false;
  std::string line;
  int line_num = 0;
  int ch;
  bool eof = false;
  while (!eof)
  {
    line.clear();
    while ((ch = fgetc(f)) != EOF && ch != '\n')
      if (ch != ' ' && ch != '\t' && ch != '\r')
        line.push_back((char)ch);
    if (ch == EOF)
    {
      eof = true;
      if (line.empty())
        break;
    }
    ++line_num;
    if (line.empty() || line[0] == '#' || line[0] == ';')
      continue;
    if (line[0] == '[')
    {
      out.emplace_back();
      out.back().type = line;
      continue;
    }
    size_t eq = line.find('=');
    if (eq == std::string::npos || out.empty())
    {
      fprintf(stderr, "Config file error line %d, could parse: %s\n", line_num, line.c_str());
      continue;
    }
    Option o;
    o.key = line.substr(0, eq);
    o.val = line.substr(eq + 1);
    out.back().options.push_back(o);
  }
  fclose(f);
  return true;
}

const char* FindOption(Section& s, const char* key)
{
  for (auto& o : s.options)
    if (o.key == key)
    {
      o.used = true;
      return o.val.c_str();
    }
  return nullptr;
}

const char* FindOptionStr(Section& s, const char* key, const char* def)
{
  const char* v = FindOption(s, key);
  if (v)
    return v;
  if (def)
    fprintf(stderr, "%s: Using default '%s'\n", key, def);
  return def;
}

const char* FindOptionStrQuiet(Section& s, const char* key, const char* def)
{
  const char* v = FindOption(s, key);
  return v ? v : def;
}

int FindOptionInt(Section& s, const char* key, int def)
{
  const char* v = FindOption(s, key);
  if (v)
    return atoi(v);
  fprintf(stderr, "%s: Using default '%d'\n", key, def);
  return def;
}

int FindOptionIntQuiet(Section& s, const char* key, int def)
{
  const char* v = FindOption(s, key);
  return v ? atoi(v) : def;
}

float FindOptionFloat(Section& s, const char* key, float def)
{
  const char* v = FindOption(s, key);
  if (v)
    return (float)atof(v);
  fprintf(stderr, "%s: Using default '%lf'\n", key, def);
  return def;
}

float FindOptionFloatQuiet(Section& s, const char* key, float def)
{
  const char* v = FindOption(s, key);
  return v ? (float)atof(v) : def;
}

void UnusedOption(Section& s)
{
  for (auto& o : s.options)
    if (!o.used)
      fprintf(stderr, "Unused field: '%s = %s'\n", o.key.c_str(), o.val.c_str());
}

ACTIVATION get_activation(const char* s)
{
  // src/activations.c:52-96
  static const struct { const char* n; ACTIVATION a; } tab[] = {
      {"logistic", LOGISTIC}, {"swish", SWISH}, {"mish", MISH}, {"gelu", GELU},
      {"normalize_channels", NORM_CHAN}, {"normalize_channels_softmax", NORM_CHAN_SOFTMAX},
      {"normalize_channels_softmax_maxval", NORM_CHAN_SOFTMAX_MAXVAL}, {"loggy", LOGGY},
      {"relu", RELU}, {"relu6", RELU6}, {"elu", ELU}, {"selu", SELU}, {"relie", RELIE},
      {"plse", PLSE}, {"hardtan", HARDTAN}, {"lhtan", LHTAN}, {"linear", LINEAR},
      {"ramp", RAMP}, {"leaky", LEAKY}, {"tanh", TANH}, {"stair", STAIR}};
  for (auto& t : tab)
    if (strcmp(s, t.n) == 0)
      return t.a;
  fprintf(stderr, "Couldn't find activation function %s, going with ReLU\n", s);
  return RELU;
}
