// Test driver for relp_amd/csrc/bigrat.hpp: reads "op a b" lines (op in + - * / c r; a, b as "num/den"), prints the exact
// result ("r": round trip through the 128-bit Rat, prints "overflow" when it does not fit).
#include <iostream>
#include <string>

#include "../../relp_amd/csrc/bigrat.hpp"

int main() {
    std::string op, a, b;
    while (std::cin >> op >> a >> b) {
        using relp::BigRat;
        BigRat x = BigRat::parse(a), y = BigRat::parse(b);
        if (op == "+") std::cout << (x + y).to_string() << "\n";
        else if (op == "-") std::cout << (x - y).to_string() << "\n";
        else if (op == "*") std::cout << (x * y).to_string() << "\n";
        else if (op == "/") std::cout << (x / y).to_string() << "\n";
        else if (op == "c") std::cout << cmp(x, y) << "\n";
        else if (op == "r") {
            try {
                std::cout << BigRat(x.to_rat()).to_string() << "\n";
            } catch (const relp::RatOverflow&) {
                std::cout << "overflow\n";
            }
        }
    }
    return 0;
}
