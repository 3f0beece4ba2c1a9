#include "settings.h"

#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <vector>

namespace OpticFlow {
namespace {

// Scans the document once and records "Path/To/Element@attr" -> value for every start tag.
bool ScanAttributes(const std::string& xml, std::map<std::string, std::string>& out)
{
    std::vector<std::string> stack;
    size_t pos = 0;
    while ((pos = xml.find('<', pos)) != std::string::npos) {
        if (xml.compare(pos, 4, "<!--") == 0) {
            const size_t end = xml.find("-->", pos + 4);
            if (end == std::string::npos) return false;
            pos = end + 3;
            continue;
        }
        const size_t close = xml.find('>', pos);
        if (close == std::string::npos) return false;
        std::string tag = xml.substr(pos + 1, close - pos - 1);
        pos = close + 1;
        if (tag.empty() || tag[0] == '?' || tag[0] == '!') continue;
        if (tag[0] == '/') {
            if (stack.empty()) return false;
            stack.pop_back();
            continue;
        }
        const bool self_closing = tag.back() == '/';
        if (self_closing) tag.pop_back();
        size_t i = 0;
        while (i < tag.size() && !isspace(static_cast<unsigned char>(tag[i]))) ++i;
        std::string path;
        for (const auto& s : stack) path += s + "/";
        path += tag.substr(0, i);
        out[path] = "";  // element seen
        while (i < tag.size()) {  // name = "value" pairs, blanks allowed around '='
            while (i < tag.size() && isspace(static_cast<unsigned char>(tag[i]))) ++i;
            size_t name_end = i;
            while (name_end < tag.size() && tag[name_end] != '=' && !isspace(static_cast<unsigned char>(tag[name_end])))
                ++name_end;
            if (name_end == i) break;
            const std::string name = tag.substr(i, name_end - i);
            i = tag.find('=', name_end);
            if (i == std::string::npos) return false;
            ++i;
            while (i < tag.size() && isspace(static_cast<unsigned char>(tag[i]))) ++i;
            if (i >= tag.size() || (tag[i] != '"' && tag[i] != '\'')) return false;
            const char quote = tag[i];
            const size_t value_end = tag.find(quote, i + 1);
            if (value_end == std::string::npos) return false;
            out[path + "@" + name] = tag.substr(i + 1, value_end - i - 1);
            i = value_end + 1;
        }
        if (!self_closing) stack.push_back(tag.substr(0, tag.find_first_of(" \t\r\n")));
    }
    return stack.empty();
}

}  // namespace

int Settings::LoadSettings(std::string fileName)
{
    std::ifstream f(fileName.c_str());
    if (!f) {
        std::cout << "Cannot read settings file: " << fileName << std::endl;
        return -1;
    }
    std::stringstream buffer;
    buffer << f.rdbuf();
    std::map<std::string, std::string> attrs;
    if (!ScanAttributes(buffer.str(), attrs) || attrs.empty()) {
        std::cout << "Problem with parsing settings file: " << fileName << std::endl;
        return -1;
    }
    // the root element's name is not checked by the reference either (settings.cpp:83-90)
    std::string root;
    for (const auto& kv : attrs)
        if (kv.first.find('/') == std::string::npos && kv.first.find('@') == std::string::npos) root = kv.first;
    bool missing = false;
    auto get = [&](const std::string& path, bool mandatory = true) -> std::string {
        auto it = attrs.find(root + "/" + path);
        if (it == attrs.end()) {
            if (mandatory) {
                std::cout << "Settings: missing " << path << std::endl;
                missing = true;
            }
            return std::string();
        }
        return it->second;
    };
    inputPath = get("Input/Path@inputPath");
    outputPath = get("Output/Path@outputPath");
    fileName1 = get("Input/Mode/Files@file1");
    fileName2 = get("Input/Mode/Files@file2");
    press_key = std::atoi(get("Parameters/Method@key").c_str()) != 0;
    width = std::atoi(get("Input/Mode@Nx").c_str());
    height = std::atoi(get("Input/Mode@Ny").c_str());
    sigma = static_cast<float>(std::atof(get("Parameters/Solver/Model@sigma").c_str()));
    iterInner = std::atoi(get("Parameters/Solver/Iterations@inner").c_str());
    iterOuter = std::atoi(get("Parameters/Solver/Iterations@outer").c_str());
    levels = std::atoi(get("Parameters/Solver/Warping@levels").c_str());
    warpScale = static_cast<float>(std::atof(get("Parameters/Solver/Warping@scaling").c_str()));
    medianRadius = std::atoi(get("Parameters/Solver/Warping@medianRadius").c_str());
    alpha = static_cast<float>(std::atof(get("Parameters/Solver/Model@alpha").c_str()));
    e_smooth = static_cast<float>(std::atof(get("Parameters/Solver/Model@e_smooth").c_str()));
    e_data = static_cast<float>(std::atof(get("Parameters/Solver/Model@e_data").c_str()));
    const std::string type = get("Input/Mode@imageType", false);
    if (!type.empty()) imageType = type;
    const std::string constancy = get("Parameters/Method@dataConstancy", false);
    if (!constancy.empty()) dataConstancy = constancy;
    return missing ? -1 : 0;
}

}  // namespace OpticFlow
