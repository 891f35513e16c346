// stand-in for <ros/ros.h> (ROS is not installed): a NodeHandle whose param() serves values from a process-wide table that
// the test fills through ref_sys_set_param (else the caller's default, as a parameter server without that key would), and
// publishers that drop what they are given.  TEST INFRASTRUCTURE ONLY.
#pragma once
#include <cstdio>
#include <map>
#include <sstream>
#include <string>
#include <vector>
namespace ros {
inline std::map<std::string, std::string> &lvba_param_table() { static std::map<std::string, std::string> t; return t; }
template <class T> struct LvbaParse {
    static bool get(const std::string &s, T &out) { std::istringstream is(s); return bool(is >> out); }
};
template <> struct LvbaParse<std::string> { static bool get(const std::string &s, std::string &out) { out = s; return true; } };
template <> struct LvbaParse<bool> {
    static bool get(const std::string &s, bool &out) { out = (s == "1" || s == "true" || s == "True"); return true; }
};
template <class E> struct LvbaParse<std::vector<E>> {
    static bool get(const std::string &s, std::vector<E> &out)
    {
        out.clear();
        std::string u = s;
        for (char &c : u) if (c == ',' || c == '[' || c == ']') c = ' ';
        std::istringstream is(u);
        E e;
        while (is >> e) out.push_back(e);
        return true;
    }
};
struct Time { static Time now() { return Time(); } };
class Publisher { public: template <class M> void publish(const M &) const {} };
class NodeHandle {
  public:
    template <class T> bool param(const std::string &key, T &var, const T &dflt) const
    {
        const T d = dflt; // the reference passes the variable itself as its own default in places
        auto it = lvba_param_table().find(key);
        if (it != lvba_param_table().end() && LvbaParse<T>::get(it->second, var)) return true;
        var = d;
        return false;
    }
    template <class M> Publisher advertise(const std::string &, int, bool = false) { return Publisher(); }
    bool ok() const { return true; }
};
inline void spin() {}
} // namespace ros
#define ROS_WARN(...) do { std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); } while (0)
#define ROS_INFO(...) do { std::printf(__VA_ARGS__); std::printf("\n"); } while (0)
#define ROS_ERROR(...) do { std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); } while (0)
